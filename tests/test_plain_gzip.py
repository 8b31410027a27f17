"""Plain (non-BGZF) gzip input of read_bcf: htslib reads such a file through zlib (bgzf.c:828-893, check_header :896-905); here the members are
inflated by the serial device decoder (csrc/gzip_serial.hip) and what they hold is read as the uncompressed text would be.  The checker is the
oracle's text reader on the bytes Python's zlib gives for the same input (test infrastructure only)."""
import gzip
import os
import random
import struct
import sys
import zlib

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orc
import vcf_text_cases as V

CASES = dict(V.all_cases())


def _text(n, seed=1, width=1):
    rnd = random.Random(seed)
    hdr = ["##fileformat=VCFv4.2", "##contig=<ID=chr1,length=248956422>", '##INFO=<ID=DP,Number=1,Type=Integer,Description="d">', '##INFO=<ID=AF,Number=A,Type=Float,Description="d">',
           '##INFO=<ID=NOTE,Number=1,Type=String,Description="d">', "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    out = ["\n".join(hdr)]
    pos = 0
    for r in range(n):
        pos += rnd.randrange(1, 50)
        note = "".join(rnd.choice("ACGTNacgtn_-") for _ in range(rnd.randrange(0, 40 * width)))
        out.append("chr1\t%d\trs%d\t%s\t%s\t%d\tPASS\tDP=%d;AF=%.4f;NOTE=%s" % (pos, r, rnd.choice("ACGT"), rnd.choice("ACGT"), rnd.randrange(100), rnd.randrange(5000), rnd.random(), note))
    return ("\n".join(out) + "\n").encode()


def _expect(data, text_delivered, error):
    """rows of the lines that are whole inside the bytes the reference's reader is handed; an error behind them when the stream failed"""
    import duckhts_amd
    if error:
        cut = text_delivered.rfind(b"\n") + 1
        text_delivered = text_delivered[:cut]
    exp = orc.bcf_read(text_delivered)
    got = duckhts_amd.read_bcf(data)
    d = orc.bcf_cols_diff(exp, got)
    assert d is None, d
    assert exp["status"] == 0
    assert (got["status"] < 0) == bool(error), (got["status"], error)
    return got


def _member(raw, level=6, flags=0, extra=b"", name=b"", comment=b""):
    """a gzip member with chosen header fields (RFC 1952)"""
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(raw) + co.flush()
    h = b"\x1f\x8b\x08" + bytes([flags]) + b"\0\0\0\0\x00\x03"
    if flags & 4:
        h += struct.pack("<H", len(extra)) + extra
    if flags & 8:
        h += name + b"\0"
    if flags & 16:
        h += comment + b"\0"
    if flags & 2:
        h += struct.pack("<H", zlib.crc32(h) & 0xffff)
    return h + body + struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw) & 0xffffffff)


@pytest.mark.gpu
def test_gpu_plain_gzip_members():
    """one member at every kind of DEFLATE block (stored, fixed, dynamic), header fields (FEXTRA that is not BGZF's, FNAME, FCOMMENT, FHCRC), several
    members in a row, matches that reach 32 KiB back, a file of some size"""
    small = CASES["numbers_plain"]
    for level in (0, 1, 6, 9):
        _expect(gzip.compress(small, level), small, False)
    big = _text(60000, seed=2)
    assert len(big) > 3 << 20
    for level in (0, 1, 9):
        got = _expect(gzip.compress(big, level), big, False)
        assert got["n_rows"] == 60000
    t = _text(300, seed=3)
    for flags, kw in ((8, dict(name=b"x.vcf")), (4, dict(extra=b"AB\x02\x00\x01\x02")), (4 | 8 | 16 | 2, dict(extra=b"ZZ\x00\x00", name=b"n", comment=b"c c")), (16, dict(comment=b"hello"))):
        _expect(_member(t, flags=flags, **kw), t, False)
    # members in a row: the second one's matches do not reach into the first one's bytes, the text simply goes on
    a, b, c3 = t[:5000], t[5000:40000], t[40000:]
    _expect(gzip.compress(a) + _member(b, level=0) + gzip.compress(c3, 9), t, False)
    # long repeats: distances up to the whole window
    rep = _text(2000, seed=4, width=30)
    twice = rep + rep[rep.index(b"\nchr1\t") + 1:]
    _expect(gzip.compress(twice, 9), twice, False)


@pytest.mark.gpu
def test_gpu_plain_gzip_streams_that_fail():
    """htslib hands the text out in chunks of 64 KiB and a call that meets an error returns nothing (inflate_gzip_block, bgzf.c:828-893): the rows
    are those of the lines that are whole in front of the last chunk boundary before the error, and the scan ends in an error"""
    big = _text(20000, seed=5)
    z = gzip.compress(big, 6)

    def delivered(data, pos_err):
        return big[: (pos_err // 65536) * 65536]
    # the file ends inside the member
    for cut in (len(z) - 3, len(z) - 9, len(z) // 2, len(z) // 3 + 1):
        d = zlib.decompressobj(31)
        n_out = len(d.decompress(z[:cut]))
        _expect(z[:cut], delivered(z, n_out), True)
    import duckhts_amd
    with pytest.raises(duckhts_amd.DhtsError):                            # nothing of the text comes out: the file cannot be told to be VCF (hts_open fails in the reference)
        duckhts_amd.read_bcf(z[:40])
    # a CRC / a length that does not match: zlib says so at the member's end
    for at in (len(z) - 8, len(z) - 2):
        bad = bytearray(z); bad[at] ^= 0x10
        _expect(bytes(bad), delivered(z, len(big)), True)
    # bytes behind the member that are not a member
    _expect(z + b"garbage!", delivered(z, len(big)), True)
    # damage inside the stream: whatever zlib makes of it -- an error somewhere, or a CRC mismatch at the end
    rnd = random.Random(9)
    for _ in range(6):
        bad = bytearray(z); at = rnd.randrange(100, len(z) - 10); bad[at] ^= 1 << rnd.randrange(8)
        d = zlib.decompressobj(31); out = b""
        try:
            out = d.decompress(bytes(bad))
            err = not d.eof
        except zlib.error:
            # the bytes zlib had produced when it met the error: decode again a piece at a time
            d = zlib.decompressobj(31); out = b""; err = True
            for i in range(0, len(bad), 64):
                try:
                    out += d.decompress(bytes(bad[i:i + 64]))
                except zlib.error:
                    break
        if not err and out == big:
            continue
        import duckhts_amd
        got = duckhts_amd.read_bcf(bytes(bad))
        assert got["status"] < 0
        # rows: never more than the lines in front of the error (the exact byte zlib stops at inside a 64-byte piece is not reproduced here)
        whole = out[: (len(out) // 65536) * 65536]
        lo = orc.bcf_read(whole[: whole.rfind(b"\n") + 1])["n_rows"] if b"\n" in whole else 0
        assert got["n_rows"] <= lo + 1 and got["n_rows"] >= max(0, lo - 1200), (got["n_rows"], lo)


@pytest.mark.gpu
def test_gpu_plain_gzip_through_the_table_function(tmp_path):
    """read_bcf('file.vcf.gz') where the file was written by gzip, not bgzip: bind reads the header from the head of the file, the scan the rows"""
    import subprocess
    import duckhts_amd
    host = os.path.join(os.path.dirname(os.path.abspath(__file__)), "minihost", "minihost")
    text = _text(50000, seed=7)
    p = tmp_path / "plain.vcf.gz"
    p.write_bytes(gzip.compress(text, 6))
    r = subprocess.run([host, duckhts_amd.LIB_PATH, "read_bcf", str(p), "-t", "1", "-r", "2", "-p", "0,1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK rows=50000" in r.stdout, r.stdout[-400:]
    # a header longer than the first head the bind stages (1 MiB of compressed bytes inflate to more than the header needs; a 3 MB header does not fit)
    rnd = random.Random(11); alphabet = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
    long_hdr = ("##fileformat=VCFv4.2\n" + "".join('##note%05d=%s\n' % (i, "".join(rnd.choice(alphabet) for _ in range(150))) for i in range(20000)) + '##INFO=<ID=K00007,Number=1,Type=Integer,Description="k">\n' +
                "##contig=<ID=chr1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\nchr1\t5\t.\tA\tC\t.\t.\tK00007=3\n").encode()
    p2 = tmp_path / "longhdr.vcf.gz"
    p2.write_bytes(gzip.compress(long_hdr, 1))
    assert os.path.getsize(p2) > (1 << 20)
    r = subprocess.run([host, duckhts_amd.LIB_PATH, "read_bcf", str(p2), "-t", "1", "-r", "1", "-p", "0,1"], capture_output=True, text=True)
    assert r.returncode == 0 and "OK rows=1 " in r.stdout, r.stdout[-400:] + r.stderr[-400:]


@pytest.mark.gpu
def test_gpu_bgunzip_of_plain_gzip(tmp_path):
    """bgunzip reads what bgzf_read reads: a plain gzip file of any content comes back as its bytes; one that fails is a read error and leaves no output"""
    import duckhts_amd
    rnd = random.Random(3)
    raw = bytes(rnd.randrange(256) for _ in range(70000)) + b"some text that repeats, some text that repeats\n" * 4000 + bytes(300000)
    src, back = str(tmp_path / "any.gz"), str(tmp_path / "any.back")
    ctx = duckhts_amd.Context(0)
    try:
        for level in (0, 6):
            z = gzip.compress(raw, level)
            open(src, "wb").write(z)
            assert ctx.bgunzip_file(src, back) == (len(z), len(raw)) and open(back, "rb").read() == raw
        open(src, "wb").write(gzip.compress(raw, 6)[:-4000])
        with pytest.raises(duckhts_amd.DhtsError, match="read error"):
            ctx.bgunzip_file(src, back + "2")
        assert not os.path.exists(back + "2")
    finally:
        ctx.close()
