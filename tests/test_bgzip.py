"""bgzip / bgunzip on the device (src/bgzip.c -> bgzf_write / bgzf_read).

No two DEFLATE implementations produce the same bytes, so the parity statement for a compressor is: every reader gives the input back
(python's gzip = zlib, the oracle's inflate, this library's own read path), the container is BGZF as bgzf.c writes it (one member per 0xff00
input bytes, BC extra field with the right BSIZE, CRC-32, ISIZE, the 28-byte EOF block), and the table functions return the reference's row."""
import gzip
import os
import random
import struct
import zlib

import numpy as np
import pytest

import orc
import vcf_text_cases as V
import vep_cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EOF_BLOCK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def bgzf_members(data):
    """[(payload bytes, crc, isize)] of a BGZF file; asserts the container fields bgzf.c writes (bgzf.c:509-620)"""
    out, p = [], 0
    while p < len(data):
        assert data[p:p + 4] == b"\x1f\x8b\x08\x04" and data[p + 10:p + 16] == b"\x06\x00BC\x02\x00", p
        bsize = struct.unpack_from("<H", data, p + 16)[0] + 1
        assert 26 <= bsize <= 65536 and p + bsize <= len(data)
        crc, isize = struct.unpack_from("<II", data, p + bsize - 8)
        out.append((data[p + 18:p + bsize - 8], crc, isize))
        p += bsize
    return out


def check_file(raw, z):
    assert z[-28:] == EOF_BLOCK
    ms = bgzf_members(z)
    assert len(ms) == (len(raw) + 65279) // 65280 + 1
    at = 0
    for pay, crc, isize in ms[:-1]:
        piece = zlib.decompress(pay, -15)
        assert piece == raw[at:at + 65280] and isize == len(piece) and crc == zlib.crc32(piece)
        at += isize
    assert at == len(raw) and gzip.decompress(z) == raw


def inputs():
    rnd = random.Random(11)
    text = vep_cases.fixture_text().encode()
    bam = open(os.path.join(GOLD, "range.bam"), "rb").read()
    cases = {
        "empty": b"", "one_byte": b"x", "three_bytes": b"abc", "four_equal": b"aaaa", "short_run": b"a" * 300, "zeros_block": bytes(65280), "zeros_block_plus_one": bytes(65281),
        "just_under_a_block": bytes(rnd.getrandbits(8) for _ in range(65279)), "random_3_blocks": rnd.randbytes(3 * 65280 + 17),
        "vcf_text": text, "bam_inflated": gzip.decompress(bam), "period_7": b"ACGTTGA" * 40000, "period_40000": (rnd.randbytes(40000)) * 4,
        "far_matches": b"".join(rnd.choice([b"the quick brown fox ", b"jumps over ", b"the lazy dog ", rnd.randbytes(3)]) for _ in range(60000)),
        "high_bytes": bytes(rnd.choice(b"\xf0\xf1\xf2\xff\x90") for _ in range(200000)), "match_at_the_end": b"xyz" * 5 + rnd.randbytes(65280 - 30) + b"xyz" * 5,
    }
    return cases


@pytest.mark.gpu
@pytest.mark.parametrize("level", [-1, 0, 6])
def test_gpu_bgzf_compress_round_trips(level):
    import duckhts_amd
    ctx = duckhts_amd.Context(0)
    try:
        for name, raw in inputs().items():
            z = ctx.bgzf_compress(raw, level)
            check_file(raw, z)
            if level == 0:
                assert len(z) == len(raw) + 31 * ((len(raw) + 65279) // 65280) + 28, name
        # the oracle's reader and this library's own read path
        raw = inputs()["vcf_text"]
        z = ctx.bgzf_compress(raw, level)
        assert orc.bcf_cols_diff(orc.bcf_read(z), orc.bcf_read(raw)) is None
        got = duckhts_amd.read_bcf(z)
        assert orc.bcf_cols_diff(orc.bcf_read(raw), got) is None and got["n_rows"] == 802
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_compression_is_worth_having():
    """sizes against zlib on the same 0xff00-byte pieces: within a factor of level 1 on text and binary, far below stored"""
    import duckhts_amd
    ctx = duckhts_amd.Context(0)
    try:
        for name in ("vcf_text", "bam_inflated", "period_7", "far_matches"):
            raw = inputs()[name]
            z = ctx.bgzf_compress(raw)
            z1 = sum(len(zlib.compress(raw[k:k + 65280], 1)) for k in range(0, len(raw), 65280))
            print(f"bgzip size {name}: raw {len(raw)}, device {len(z)}, zlib level 1 {z1}, level 6 {sum(len(zlib.compress(raw[k:k + 65280], 6)) for k in range(0, len(raw), 65280))}")
            assert len(z) < 0.75 * len(raw) and (len(z) < 1.6 * z1 or len(z) < 0.02 * len(raw)), (name, len(raw), len(z), z1)
        raw = inputs()["random_3_blocks"]
        assert len(raw) < len(ctx.bgzf_compress(raw)) <= len(raw) + 31 * 4 + 28         # incompressible: stored blocks (31 bytes of framing each)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_bgzip_and_bgunzip_files(tmp_path):
    import duckhts_amd
    raw = V.text([V.L(pos=i + 1, info="DP=%d" % (i % 90)) for i in range(300000)])
    src, dst, back = (os.path.join(str(tmp_path), n) for n in ("a.vcf", "a.vcf.gz", "a.back"))
    open(src, "wb").write(raw)
    ctx = duckhts_amd.Context(0)
    try:
        nin, nout = ctx.bgzip_file(src, dst)
        z = open(dst, "rb").read()
        assert (nin, nout) == (len(raw), len(z))
        check_file(raw, z)
        assert ctx.bgunzip_file(dst, back) == (len(z), len(raw)) and open(back, "rb").read() == raw
        # a file htslib wrote
        assert ctx.bgunzip_file(os.path.join(GOLD, "vcf_file.bcf"), back)[1] == len(gzip.decompress(open(os.path.join(GOLD, "vcf_file.bcf"), "rb").read()))
        assert open(back, "rb").read() == gzip.decompress(open(os.path.join(GOLD, "vcf_file.bcf"), "rb").read())
        # damage: a flipped payload bit is a read error, nothing silently short
        bad = bytearray(z); bad[len(bad) // 2] ^= 0x10
        open(dst, "wb").write(bytes(bad))
        with pytest.raises(duckhts_amd.DhtsError, match="read error"):
            ctx.bgunzip_file(dst, back)
        assert not os.path.exists(back)                                               # no partial output behind an error
        one = os.path.join(str(tmp_path), "one.byte"); open(one, "wb").write(b"\x1f")
        assert ctx.bgunzip_file(one, back) == (1, 1) and open(back, "rb").read() == b"\x1f"      # shorter than the gzip magic: not gzip, handed through
        assert ctx.bgunzip_file(src, back) == (len(raw), len(raw)) and open(back, "rb").read() == raw      # not gzip: handed through (bgzf_open reads it transparently)
        open(dst, "wb").write(gzip.compress(raw[:5000]))
        assert ctx.bgunzip_file(dst, back)[1] == 5000 and open(back, "rb").read() == raw[:5000]      # plain gzip: inflated, as htslib does through zlib (round 4; tests/test_plain_gzip.py)
        with pytest.raises(duckhts_amd.DhtsError, match="cannot open input"):
            ctx.bgzip_file(src + ".nope", dst)
    finally:
        ctx.close()
    # the compressed text reads like the plain text, and can be indexed and queried
    import test_vcf_region as TR
    exp = orc.bcf_read(raw)
    got = duckhts_amd.read_bcf(z, max_blocks=16)
    assert orc.bcf_cols_diff(exp, got) is None
    tbi = TR.build_index(z, 0)[1]
    assert TR.region_check(z, "chr1:150000-150009", tbi) == 10
