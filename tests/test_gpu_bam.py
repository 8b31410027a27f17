"""GPU parity tests proper: the HIP scan path (through the C ABI) vs the oracle, bit-exact."""
import struct

import numpy as np
import pytest

import cases
import duckhts_amd
import orc
import os

from conftest import GOLDEN, ROOT, read_golden
from duckhts_amd import synth

pytestmark = pytest.mark.gpu

COLS = duckhts_amd.BAM_COLUMNS


def assert_same(got, exp, ctx=""):
    assert got["n_rows"] == exp["n_rows"], f"{ctx}: rows {got['n_rows']} != {exp['n_rows']}"
    assert (got["status"] < 0) == (exp["status"] < 0), f"{ctx}: status {got['status']} vs {exp['status']}"
    for k in COLS:
        a, b = got[k], exp[k]
        if isinstance(b, np.ndarray):
            assert np.array_equal(np.asarray(a), b), f"{ctx}: column {k} differs"
        else:
            if list(a) != list(b):
                for i, (x, y) in enumerate(zip(a, b)):
                    assert x == y, f"{ctx}: column {k} row {i}: {x!r} != {y!r}"
                raise AssertionError(f"{ctx}: column {k} length differs")


# ---- BGZF: block discovery + inflate + CRC ----

@pytest.mark.parametrize("name", ["range.bam", "bgzf_boundaries1.bam", "bgzf_boundaries3.bam", "vcf_file.bcf", "colons.bam"])
def test_bgzf_inflate_golden(name):
    d = read_golden(name)
    z = orc.bgzf_inflate_all(d)
    ctx = duckhts_amd.Context(0)
    ctx.open(d)
    nb = ctx.bgzf_index()
    assert nb == z["n_blocks"]
    coff, clen, isize, st = ctx.bgzf_table(nb)
    assert st == 0
    assert np.array_equal(coff.astype(np.int64), z["coff"]) and np.array_equal(clen.astype(np.int32), z["clen"])
    assert np.array_equal(isize.astype(np.int32), z["ulen"])
    out, bst = ctx.bgzf_inflate(0, nb, len(z["data"]))
    assert np.all(bst == 0)
    assert out.tobytes() == z["data"]
    ctx.close()


def test_packed_scratch_retry(monkeypatch):
    """phase A packs every block's literals and tokens into a pool sized for ~68 KB per block; a block that finds the pool exhausted is
    marked (DHTS_BLK_ERR_SCRATCH) and the host decodes the range again with full-size room: a pool of 2 KB per block forces that path"""
    monkeypatch.setenv("DHTS_POOL_PER_BLOCK", "2048")
    data = synth.bam_file(30000, seed=77)
    z = orc.bgzf_inflate_all(data)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data)
        nb = ctx.bgzf_index()
        out, bst = ctx.bgzf_inflate(0, nb, len(z["data"]))
        assert np.all(bst == 0) and out.tobytes() == z["data"]
    finally:
        ctx.close()
    got = duckhts_amd.read_bam(data, device=0)
    exp = orc.bam_read(data)
    assert got["n_rows"] == exp["n_rows"] == 30000
    for k in duckhts_amd.BAM_COLUMNS:
        assert list(got[k]) == list(exp[k]), k


@pytest.mark.parametrize("pool_per_block", [None, "2048"])
def test_damaged_block_in_the_first_batch_of_a_speculative_shard(monkeypatch, pool_per_block):
    """the batch driver queues the tile pass behind the inflate before the host has seen the blocks' status: a damaged block inside the
    FIRST batch of a shard that starts mid-stream (its first record found by speculation, possibly walking bytes of the failed block) must
    end that shard's rows in front of the damage, exactly where the sequential scan of the damaged file ends -- also when the packed
    phase-A scratch overflows at the same time (DHTS_POOL_PER_BLOCK)"""
    if pool_per_block:
        monkeypatch.setenv("DHTS_POOL_PER_BLOCK", pool_per_block)
    data = bytearray(cases.case_basic(payload=777, n=3000, seed=12))
    p, blocks = 0, []
    while p + 18 <= len(data):
        bl = (data[p + 16] | (data[p + 17] << 8)) + 1
        blocks.append((p, bl)); p += bl
    nb = len(blocks)
    for k in (nb * 5 // 8, nb // 2 + 1):
        d = bytearray(data)
        o, bl = blocks[k]
        d[o + 18 + (bl - 26) // 2] ^= 0x55                           # a payload byte in the middle of the block
        d = bytes(d)
        exp = orc.bam_read(d)
        assert 0 < exp["n_rows"] < 3000
        names = []
        for rank in range(2):
            ctx = duckhts_amd.Context(0)
            try:
                ctx.open(d); ctx.bgzf_index(); ctx.bam_open()
                ctx.set_block_range(0 if rank == 0 else nb // 2, nb // 2 if rank == 0 else nb, rank > 0)
                status = 0
                while status == 0:
                    b = ctx.next_batch(0)                           # the whole shard in one (first) batch
                    if b.n_rows:
                        n = int(b.n_rows)
                        off = ctx.d2h(b.qname.off, n + 1, np.uint32); ln = ctx.d2h(b.qname.len, n, np.uint32); raw = ctx.d2h(b.qname.bytes, int(b.qname.nbytes), np.uint8).tobytes()
                        names += [raw[off[i]:off[i] + ln[i]] for i in range(n)]
                    status = int(b.status)
                assert (status == 1) if rank == 0 else (status < 0), (rank, status)
            finally:
                ctx.close()
        assert names == list(exp["QNAME"]), (k, len(names), exp["n_rows"])


@pytest.mark.parametrize("case", ["basic", "basic_small_blocks", "basic_tiny_blocks", "basic_stored", "basic_level1",
                                  "basic_level9", "fixed_huffman", "long_record", "empty_blocks"])
def test_bgzf_inflate_cases(case):
    d = cases.ALL_CASES[case]()
    z = orc.bgzf_inflate_all(d)
    ctx = duckhts_amd.Context(0)
    ctx.open(d)
    nb = ctx.bgzf_index()
    assert nb == z["n_blocks"]
    out, bst = ctx.bgzf_inflate(0, nb, len(z["data"]))
    assert np.all(bst == 0) and out.tobytes() == z["data"]
    # sub-ranges land at the right offsets too
    if nb > 3:
        lo = int(z["ulen"][:1].sum()); hi = int(z["ulen"][:3].sum())
        out2, _ = ctx.bgzf_inflate(1, 2, hi - lo)
        assert out2.tobytes() == z["data"][lo:hi]
    ctx.close()


def _lz_payloads():
    """raw payloads that stress the LZ77 window: far sources (read back from flushed output), ring wrap, long and overlapping copies"""
    import random
    rnd = random.Random(11)
    R = lambda n: bytes(rnd.getrandbits(8) for _ in range(n))
    a = R(32768 - 100)
    out = {
        "far_max_distance": a + a[:258] + R(50) + a[1000:1300],                   # distances just under 32 KiB
        "far_many": b"".join(R(700) for _ in range(12)) * 7,                      # 8.4 KB period: every copy is older than the 8 KiB ring
        "near_and_far": (R(5000) + b"ABCDEFGH" * 40 + R(3000)) * 6,
        "zeros": b"\0" * 65280,                                                   # dist 1, len 258 chains
        "period3": b"xyz" * 21000,
        "period_300": R(300) * 217,
        "random": R(65000),
        "text": (b"the quick brown fox jumps over the lazy dog %d\n" * 1400) % tuple(range(1400)),
        "long_literal_runs": b"".join(R(3000) + b"Q" * 600 for _ in range(17)),
    }
    return out


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("name", sorted(_lz_payloads()))
def test_bgzf_inflate_window_stress(name, level):
    import bamwriter as bw
    raw = _lz_payloads()[name]
    d = bw.bgzf_file(raw, payload=65280, level=level)
    ctx = duckhts_amd.Context(0)
    ctx.open(d)
    nb = ctx.bgzf_index()
    out, bst = ctx.bgzf_inflate(0, nb, len(raw))
    ctx.close()
    assert np.all(bst == 0) and out.tobytes() == raw


def test_bgzf_errors_flagged_per_block():
    for maker, code in ((cases.case_bad_crc, -4), (cases.case_bad_deflate, None)):
        d = maker()
        ctx = duckhts_amd.Context(0)
        ctx.open(d)
        nb = ctx.bgzf_index()
        _, clen, isize, _ = ctx.bgzf_table(nb)
        out, bst = ctx.bgzf_inflate(0, nb, int(isize.astype(np.int64).sum()))
        bad = np.nonzero(bst)[0]
        assert len(bad) == 1
        if code is not None:
            assert bst[bad[0]] == code
        z = orc.bgzf_inflate_all(d)
        assert z["n_blocks"] == bad[0]          # the oracle's stream ends exactly at that block
        ctx.close()


def test_bgzf_index_signature_inside_payload():
    """a BGZF signature embedded in a stored payload must not be mistaken for a block start"""
    import bamwriter as bw
    inner = bw.bgzf_block(b"inner block")
    raw = b"x" * 100 + inner + b"y" * 100
    d = bw.bgzf_file(raw, level=0) 
    z = orc.bgzf_inflate_all(d)
    ctx = duckhts_amd.Context(0)
    ctx.open(d)
    nb = ctx.bgzf_index()
    assert nb == z["n_blocks"] == 2
    out, bst = ctx.bgzf_inflate(0, nb, len(z["data"]))
    assert out.tobytes() == z["data"] == raw
    ctx.close()


# ---- read_bam: golden fixtures ----

@pytest.mark.parametrize("name", ["range.bam", "bgzf_boundaries1.bam", "bgzf_boundaries2.bam", "bgzf_boundaries3.bam",
                                  "no_hdr_sq_1.bam", "colons.bam"])
def test_read_bam_golden(name):
    d = read_golden(name)
    got = duckhts_amd.read_bam(d)
    exp = orc.bam_read(d)
    assert_same(got, exp, name)
    assert got["header"]["ref_names"] == exp["ref_names"]
    assert got["header"]["text"] == exp["text"]
    assert got["header"]["first_rec_uoff"] == exp["first_rec_off"]


def test_read_bam_range_sql_expectations():
    """/root/reference/test/sql/duckhts.test:129-137, straight from the GPU path"""
    got = duckhts_amd.read_bam(read_golden("range.bam"))
    assert got["n_rows"] == 112
    assert (got["QNAME"][0], int(got["FLAG"][0]), got["RNAME"][0], int(got["POS"][0]), int(got["MAPQ"][0])) == \
        (b"HS18_09653:4:1315:19857:61712", 145, b"CHROMOSOME_I", 914, 23)
    assert got["READ_GROUP_ID"][0] == b"1" and got["SAMPLE_ID"][0] == b"ERS225193"


# ---- read_bam: edge cases ----

@pytest.mark.parametrize("case", sorted(cases.ALL_CASES))
def test_read_bam_cases(case):
    d = cases.ALL_CASES[case]()
    exp = orc.bam_read(d)
    got = duckhts_amd.read_bam(d)
    assert_same(got, exp, case)


@pytest.mark.parametrize("max_blocks", [1, 2, 3, 7])
def test_read_bam_small_batches_carry(max_blocks):
    """records straddle batch boundaries: the carry logic must reproduce the single-pass result"""
    for case in ["basic_small_blocks", "quirks", "long_record", "err_cigar_qlen", "truncated_record", "empty_blocks"]:
        d = cases.ALL_CASES[case]()
        exp = orc.bam_read(d)
        got = duckhts_amd.read_bam(d, max_blocks=max_blocks)
        assert_same(got, exp, f"{case}/mb{max_blocks}")


def test_read_bam_synthetic_wgs_shape():
    from duckhts_amd import synth
    arr, st = synth.bam_segment(300000, seed=42)
    d = arr.tobytes()
    exp = orc.bam_read(d)
    got = duckhts_amd.read_bam(d)
    assert_same(got, exp, "synth300k")
    got = duckhts_amd.read_bam(d, max_blocks=100)
    assert_same(got, exp, "synth300k/mb100")


def test_read_bam_with_the_next_batch_prefetch(monkeypatch):
    """DHTS_PREFETCH=1: phase B of batch k+1 on a second stream beside the record stage of batch k (the default until round 3)"""
    from duckhts_amd import synth
    monkeypatch.setenv("DHTS_PREFETCH", "1")
    d = synth.bam_file(200000, seed=5)
    exp = orc.bam_read(d)
    for mb in (64, 100, 1000):
        assert_same(duckhts_amd.read_bam(d, max_blocks=mb), exp, f"prefetch/mb{mb}")


@pytest.mark.parametrize("world", [2, 3])
def test_read_bam_sharded_concat(world):
    """BGZF block ranges shard across ranks; concatenating the shards gives the sequential scan, and the
    hand-off offsets chain (end of shard r == first record of shard r+1)."""
    from duckhts_amd import synth
    arr, st = synth.bam_segment(120000, seed=9)
    d = arr.tobytes()
    exp = orc.bam_read(d)
    parts = [duckhts_amd.read_bam(d, shard=(r, world), max_blocks=50) for r in range(world)]
    assert sum(p["n_rows"] for p in parts) == exp["n_rows"]
    for k in COLS:
        cat = [x for p in parts for x in list(p[k])]
        assert cat == list(exp[k]), k


def test_full_size_properties():
    """size-independent properties on a larger input (no oracle pass over the whole thing):
    row count == records generated, coordinate-sortedness, mate symmetry, checksum of checksums of QUAL/SEQ lengths."""
    from duckhts_amd import synth
    n = 2_000_000
    arr, st = synth.bam_segment(n, seed=123)
    ctx = duckhts_amd.Context(0)
    ctx.open(arr)
    ctx.bgzf_index()
    hdr = ctx.bam_open()
    rows = 0; last_key = -1; status = 0; seq_bytes = 0
    while True:
        b = ctx.next_batch(4096)
        if b.n_rows:
            tid = ctx.d2h(b.tid, b.n_rows, np.int32).astype(np.int64)
            pos = ctx.d2h(b.pos, b.n_rows, np.int64)
            key = tid * (1 << 32) + pos
            assert key[0] >= last_key and np.all(np.diff(key) >= 0)
            last_key = int(key[-1])
            ls = ctx.d2h(b.seq.len, b.n_rows, np.uint32)
            assert np.all(ls == 150)
            seq_bytes += int(b.seq.nbytes)
            rows += b.n_rows
        status = b.status
        if status != 0:
            break
    assert status == 1 and rows == n and seq_bytes == 150 * n
    ctx.close()


# ---- region queries (row A11) -----------------------------------------------------------------------------------------
def _region_check(data, region, index=None, **kw):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import region_oracle
    exp = orc.bam_read(data)
    keep = region_oracle.keep_mask(exp, region)
    got = duckhts_amd.read_bam(data, region=region, index=index, **kw)
    assert got["n_rows"] == int(keep.sum()), (region, got["n_rows"], int(keep.sum()))
    for k in duckhts_amd.BAM_COLUMNS:
        e = [x for x, m in zip(exp[k], keep) if m]
        assert list(got[k]) == list(e), (region, k)
    return got["n_rows"]


@pytest.mark.gpu
def test_region_golden_counts():
    """duckhts.test:139-161, 610-618: full contig 18, sub-range 2, duplicate regions dedup, explicit index"""
    data = open(os.path.join(GOLDEN, "range.bam"), "rb").read()
    bai = open(os.path.join(GOLDEN, "range.bam.bai"), "rb").read()
    for index in (None, bai):
        assert _region_check(data, "CHROMOSOME_I", index) == 18
        assert _region_check(data, "CHROMOSOME_I:1-1000", index) == 2
        assert _region_check(data, "CHROMOSOME_I:1-1000,CHROMOSOME_I:1-1000", index) == 2
        assert _region_check(data, "CHROMOSOME_II:2,000-3,500,CHROMOSOME_I:900-1000,nosuch:1-5,CHROMOSOME_V", index) > 0
        assert _region_check(data, ".", index) == 112
        assert _region_check(data, "*", index) == 0
        _region_check(data, "CHROMOSOME_III:-1500", index)
        _region_check(data, "CHROMOSOME_IV:1k-2K,{CHROMOSOME_X}:1-10", index)
    with pytest.raises(duckhts_amd.DhtsError):
        duckhts_amd.read_bam(data, region="nosuch")


@pytest.mark.gpu
def test_region_synthetic_with_unplaced_and_batches():
    data = synth.bam_file(120000, seed=5)
    for region in ("chr1:1,000,000-2,000,000", "chr2:5000000-5100000,chr2:5050000-6000000,chrX", "chrM,*", "chr21:1-1000,chr22"):
        _region_check(data, region, max_blocks=7)


# ---- BAI writer ------------------------------------------------------------------------------------
def _parse_bai(d):
    """-> (per reference: {bin: [(u, v), ...]}, linear index list), n_no_coor"""
    assert d[:4] == b"BAI\1"
    p = 4
    n_ref = struct.unpack_from("<i", d, p)[0]; p += 4
    refs = []
    for _ in range(n_ref):
        nb = struct.unpack_from("<i", d, p)[0]; p += 4
        bins = {}
        for _ in range(nb):
            b, nc = struct.unpack_from("<Ii", d, p); p += 8
            bins[b] = [struct.unpack_from("<QQ", d, p + 16 * k) for k in range(nc)]; p += 16 * nc
        ni = struct.unpack_from("<i", d, p)[0]; p += 4
        lin = list(struct.unpack_from("<%dQ" % ni, d, p)); p += 8 * ni
        refs.append((bins, lin))
    nnc = struct.unpack_from("<Q", d, p)[0] if p + 8 <= len(d) else None
    return refs, nnc


@pytest.mark.gpu
def test_bai_writer_matches_golden_index():
    """the index samtools wrote for the reference's range.bam (test/data/range.bam.bai): same bins, chunks, linear index, counts"""
    data = read_golden("range.bam")
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); ctx.bgzf_index(); ctx.bam_open()
        bai = ctx.build_index()
    finally:
        ctx.close()
    got, exp = _parse_bai(bai), _parse_bai(read_golden("range.bam.bai"))
    assert got[1] == exp[1] == 0
    assert len(got[0]) == len(exp[0]) == 7
    for t, ((gb, gl), (eb, el)) in enumerate(zip(got[0], exp[0])):
        assert gb == eb, (t, gb, eb)
        assert gl == el, (t, gl, el)
    # and the reader side accepts it: same rows as with the golden index
    for region in ("CHROMOSOME_I:1-1000", "CHROMOSOME_II:2,000-3,500,CHROMOSOME_IV", "CHROMOSOME_V"):
        a = duckhts_amd.read_bam(data, region=region, index=bai)
        b = duckhts_amd.read_bam(data, region=region, index=read_golden("range.bam.bai"))
        assert a["n_rows"] == b["n_rows"] and a["QNAME"] == b["QNAME"]


@pytest.mark.gpu
def test_bai_writer_synthetic_index_drives_region_queries():
    """a BAI built for a larger multi-batch file narrows region scans without changing their rows (vs the unindexed predicate)"""
    data = synth.bam_file(150000, seed=13)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); nb = ctx.bgzf_index(); ctx.bam_open()
        bai = ctx.build_index()
    finally:
        ctx.close()
    refs, nnc = _parse_bai(bai)
    assert nnc is not None and sum(len(r[0]) for r in refs) > 0
    for region in ("chr1:1,000,000-2,000,000", "chr2:5000000-5100000,chrX:1-50,000,000", "chr21"):
        a = duckhts_amd.read_bam(data, region=region, index=bai, max_blocks=6)
        b = duckhts_amd.read_bam(data, region=region, max_blocks=6)
        assert a["n_rows"] == b["n_rows"] and a["QNAME"] == b["QNAME"] and list(a["POS"]) == list(b["POS"])


# ---- string pass: read lengths around the chunk-map limits ---------------------------------------------------------------
@pytest.mark.gpu
def test_string_pass_mixed_read_lengths():
    """SEQ/QUAL lengths 0..1500 incl. 511/512/513 (whole-wave streaming starts above 512), '*' QUAL, QUAL bytes that become NUL"""
    import random
    import bamwriter as bw
    rng = random.Random(23)
    lens = [0, 1, 15, 16, 17, 31, 32, 33, 255, 256, 257, 511, 512, 513, 514, 1023, 1024, 1025, 1500] + [rng.randrange(0, 1500) for _ in range(400)]
    recs = []
    for i, n in enumerate(lens):
        seq = "".join(rng.choice("ACGTN") for _ in range(n)) if n else "*"
        if n and i % 7 == 3:
            qual = None                                        # 0xff fill => '*'
        elif n:
            qual = "".join(chr(33 + rng.randrange(0, 60)) for _ in range(n))
        else:
            qual = None
        recs.append(bw.record(qname=f"q{i}", flag=4, tid=-1, pos=-1, seq=seq, qual=qual))
    data = bw.bam_bytes([("ref", 100000)], recs, text="@HD\tVN:1.6\n@SQ\tSN:ref\tLN:100000\n", payload=20000, level=6)
    exp = orc.bam_read(data)
    for mb in (0, 2):
        got = duckhts_amd.read_bam(data, max_blocks=mb)
        assert got["n_rows"] == exp["n_rows"] == len(lens)
        for k in ("QNAME", "CIGAR", "SEQ", "QUAL"):
            assert got[k] == exp[k], k


# ---- edge cases of the index writer and the join ---------------------------------------------------------------------------
@pytest.mark.gpu
def test_index_writer_and_join_edge_cases():
    hdr_only = cases.case_header_only()
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(hdr_only); ctx.bgzf_index(); ctx.bam_open()
        refs, nnc = _parse_bai(ctx.build_index())
        assert nnc == 0 and all(not b and not l for b, l in refs)
    finally:
        ctx.close()
    got = duckhts_amd.read_bam(hdr_only, overlap=([0], [0], [100]))
    assert got["n_rows"] == 0 and got["OVERLAPS"] == []
    # unplaced reads only: no bins, n_no_coor counts them (hts_idx_push: tid < 0)
    un = cases.case_no_refs_unmapped()
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(un); ctx.bgzf_index(); ctx.bam_open()
        refs, nnc = _parse_bai(ctx.build_index())
        assert refs == [] and nnc == 10
    finally:
        ctx.close()
    # an unsorted file cannot be indexed (hts_idx_push: "Unsorted positions" / "Chromosome blocks not continuous")
    import bamwriter as bw
    bad = bw.bam_bytes([("a", 1000), ("b", 1000)], [bw.record(qname="x", tid=0, pos=500, cigar="4M", seq="ACGT"), bw.record(qname="y", tid=0, pos=100, cigar="4M", seq="ACGT")],
                       text="@HD\tVN:1.6\n@SQ\tSN:a\tLN:1000\n@SQ\tSN:b\tLN:1000\n")
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(bad); ctx.bgzf_index(); ctx.bam_open()
        with pytest.raises(duckhts_amd.DhtsError, match="Unsorted"):
            ctx.build_index()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_index_writer_follows_hts_idx_push_on_pos_zero_and_on_understated_lengths():
    """(a) hts_idx_push remembers the CLAMPED begin of a placed record (POS 0 = begin -1 is stored as 0) and compares it with the next
    record's unclamped begin: two placed records with POS 0 in a row are "Unsorted positions" (hts.c:2591, 2620-2633); one followed by POS 1
    is fine.  (b) htslib grows a sequence's linear index on demand: a header whose LN understates the sequence does not stop the build,
    and the index equals the one built with the right length."""
    import bamwriter as bw
    hdr = "@HD\tVN:1.6\n@SQ\tSN:a\tLN:1000\n"
    two_zero = bw.bam_bytes([("a", 1000)], [bw.record(qname="x", tid=0, pos=-1, cigar="4M", seq="ACGT"), bw.record(qname="y", tid=0, pos=-1, cigar="4M", seq="ACGT")], text=hdr)
    zero_one = bw.bam_bytes([("a", 1000)], [bw.record(qname="x", tid=0, pos=-1, cigar="4M", seq="ACGT"), bw.record(qname="y", tid=0, pos=0, cigar="4M", seq="ACGT")], text=hdr)
    for data, ok in ((two_zero, False), (zero_one, True)):
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(data); ctx.bgzf_index(); ctx.bam_open()
            if ok:
                refs, nnc = _parse_bai(ctx.build_index())
                assert nnc == 0 and len(refs) == 1
            else:
                with pytest.raises(duckhts_amd.DhtsError, match="Unsorted"):
                    ctx.build_index()
        finally:
            ctx.close()
    recs = [bw.record(qname=f"r{i}", tid=0, pos=p, cigar="50M", seq="A" * 50) for i, p in enumerate((10, 900, 5_000_000, 5_000_020, 40_000_000))]
    built = []
    for ln in (1000, 50_000_000):
        # (the same header text and stored blocks: the two files differ in the four bytes of l_ref only, so their virtual offsets are equal)
        data = bw.bam_bytes([("a", ln)], recs, text="@HD\tVN:1.6\n@SQ\tSN:a\tLN:50000000\n", level=0)
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(data); ctx.bgzf_index(); ctx.bam_open()
            built.append(_parse_bai(ctx.build_index()))
            # the scan is usable afterwards (rewound on every exit)
            assert ctx.next_batch(0).n_rows == 5
        finally:
            ctx.close()
    assert built[0] == built[1] and len(built[0][0][0][1]) >= 40_000_000 >> 14


# ---- a wrong ISIZE is not an error to htslib (it never reads the field): the block table is corrected from the decoded lengths ----
@pytest.mark.gpu
@pytest.mark.parametrize("bogus", [70000, 5, 0x04000000, 0x80000010, 0xFFFFFFFF, -1])
def test_wrong_isize_with_a_good_crc_reads_like_the_clean_file(bogus):
    """htslib never reads the ISIZE trailer field (bgzf.c:793-801 checks the CRC-32 of whatever the block inflates to): a file whose ISIZE
    is wrong is valid to the reference.  The scan places blocks by ISIZE, so after phase A the table is corrected from the decoded lengths
    (isize_repair) and the file reads like the clean one"""
    data = bytearray(cases.case_basic(payload=777, n=200, seed=2))
    clean = orc.bam_read(bytes(data))
    # blocks; damage the ISIZE field of one in the middle
    p, blocks = 0, []
    while p + 18 <= len(data) and data[p:p + 4] == b"\x1f\x8b\x08\x04":
        bl = struct.unpack_from("<H", data, p + 16)[0] + 1
        blocks.append((p, bl)); p += bl
    k = len(blocks) // 2
    at = blocks[k][0] + blocks[k][1] - 4
    if bogus == -1:                                          # 0xFFFF0000 + true length: used to wrap the 32-bit partial sums of the uoff scan
        bogus = 0xFFFF0000 + struct.unpack_from("<I", data, at)[0]
    struct.pack_into("<I", data, at, bogus)
    assert orc.bam_read(bytes(data))["n_rows"] == clean["n_rows"]          # the oracle, like htslib, does not look at ISIZE
    for mb in (0, 3):
        got = duckhts_amd.read_bam(bytes(data), max_blocks=mb)
        assert got["status"] >= 0 and got["n_rows"] == clean["n_rows"], (mb, got["status"], got["n_rows"])
        for c in COLS:
            assert list(got[c]) == list(clean[c]), c


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["first_record_block", "middle"])
def test_an_isize_of_zero_on_a_data_block_reads_like_the_clean_file(which):
    """a block that declares no bytes takes no room in the inflated stream until the table is corrected: when it is the block that holds the
    first record, the jump over the header blocks used to jump over it as well (round-2 soak seed 2001161); with the table corrected from
    the decoded lengths the file reads like the clean one (htslib never reads ISIZE)"""
    data = bytearray(cases.case_basic(payload=65280 if which == "first_record_block" else 777, level=1, n=50 if which == "first_record_block" else 200, seed=2001161))
    clean = orc.bam_read(bytes(data))
    p, blocks = 0, []
    while p + 18 <= len(data) and data[p:p + 4] == b"\x1f\x8b\x08\x04":
        bl = struct.unpack_from("<H", data, p + 16)[0] + 1
        blocks.append((p, bl)); p += bl
    k = 1 if which == "first_record_block" else len(blocks) // 2
    struct.pack_into("<I", data, blocks[k][0] + blocks[k][1] - 4, 0)
    for mb in (0, 1, 2):
        got = duckhts_amd.read_bam(bytes(data), max_blocks=mb)
        assert got["status"] >= 0 and got["n_rows"] == clean["n_rows"], (mb, got["status"], got["n_rows"])
        for c in COLS:
            assert list(got[c]) == list(clean[c]), c


@pytest.mark.gpu
def test_packed_seq_and_overlapped_fetch_at_the_c_abi():
    """dhts_bam_set_seq_packed: the batch carries 4-bit codes, len = bases, "*" = 0 bases; dhts_bam_batch_fetch_begin / _wait deliver the
    same bytes as dhts_bam_batch_fetch while the next batch is already being scanned"""
    import ctypes as C
    data = cases.case_basic(payload=777, n=3000, seed=12)
    for extra in (cases.case_error_midfile("cigar_qlen"), read_golden("range.bam")):
        exp = orc.bam_read(extra)
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(extra); ctx.bgzf_index(); hdr = ctx.bam_open(); ctx.set_seq_packed(True)
            b = ctx.next_batch(0)
            assert b.seq_packed == 1
            got = ctx.batch_to_host(b, hdr)
            assert got["SEQ"] == list(exp["SEQ"][:b.n_rows]) and got["QUAL"] == list(exp["QUAL"][:b.n_rows])
        finally:
            ctx.close()
    exp = orc.bam_read(data)
    L = duckhts_amd.lib()
    L.dhts_bam_batch_host_bytes.restype = C.c_uint64; L.dhts_bam_batch_host_bytes.argtypes = [C.c_void_p, C.c_uint32]
    L.dhts_host_alloc.restype = C.c_void_p; L.dhts_host_alloc.argtypes = [C.c_uint64]; L.dhts_host_free.argtypes = [C.c_void_p]
    L.dhts_bam_batch_fetch_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    L.dhts_bam_batch_fetch_wait.argtypes = [C.c_void_p, C.c_int]
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); ctx.bgzf_index(); ctx.bam_open(); ctx.set_seq_packed(True)
        arenas, hosts, slot, rows, seqs = [], [], 0, 0, []
        pending = None
        while True:
            b = ctx.next_batch(3)
            if b.n_rows:
                need = L.dhts_bam_batch_host_bytes(C.byref(b), 0x1FFF)
                arena = L.dhts_host_alloc(need); arenas.append(arena)
                hb = duckhts_amd.BamBatch()
                assert L.dhts_bam_batch_fetch_begin(ctx.h, C.byref(b), 0x1FFF, arena, need, C.byref(hb), slot) == 0, ctx.L.dhts_error(ctx.h)
                if pending is not None:
                    assert L.dhts_bam_batch_fetch_wait(ctx.h, pending[1]) == 0
                    hosts.append(pending[0])
                pending = (hb, slot); slot ^= 1
            if b.status != 0:
                break
        if pending is not None:
            assert L.dhts_bam_batch_fetch_wait(ctx.h, pending[1]) == 0
            hosts.append(pending[0])
        lut = b"=ACMGRSVTWYHKDBN"
        for hb in hosts:
            n = hb.n_rows
            off = np.ctypeslib.as_array(C.cast(hb.seq.off, C.POINTER(C.c_uint32)), (n + 1,)); ln = np.ctypeslib.as_array(C.cast(hb.seq.len, C.POINTER(C.c_uint32)), (n,))
            raw = np.ctypeslib.as_array(C.cast(hb.seq.bytes, C.POINTER(C.c_uint8)), (int(hb.seq.nbytes),))
            pos = np.ctypeslib.as_array(C.cast(hb.pos, C.POINTER(C.c_int64)), (n,))
            assert list(pos) == list(exp["POS"][rows:rows + n])
            for i in range(n):
                s_ = bytes(lut[(raw[off[i] + (k >> 1)] >> (0 if k & 1 else 4)) & 15] for k in range(int(ln[i]))) or b"*"
                seqs.append(s_)
            rows += n
        assert rows == exp["n_rows"] and seqs == list(exp["SEQ"])
        for a in arenas:
            L.dhts_host_free(a)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("alphabet, bits", [(b"#,:", 2), (b"!\"#$%&'()*+,-./0", 4), (None, 0)])
def test_qual_over_pcie_as_codes_of_the_batch_alphabet(alphabet, bits):
    """dhts_bam_set_qual_packed: the read-back (dhts_bam_batch_fetch and _fetch_begin / _wait) looks at which characters the batch's QUAL heap holds
    and ships 2 bits per character for at most 4 different ones, 4 bits for at most 16, the characters themselves otherwise; decoded on the host,
    the strings are those of the oracle (rows whose QUAL is "*" included: the star is a character of the heap, the fourth / sixteenth here)"""
    import ctypes as C
    import random
    import bamwriter as bw
    rnd = random.Random(7)
    recs = []
    for i in range(3000):
        n = rnd.choice([0, 1, 37, 100, 150, 151])
        if alphabet is None:
            q = bytes(rnd.randrange(33, 127) for _ in range(n)).decode()
        else:
            q = bytes(rnd.choice(alphabet) for _ in range(n)).decode()
        seq = "".join(rnd.choice("ACGT") for _ in range(n)) if n else "*"
        recs.append(bw.record(qname=f"r{i}", tid=0, pos=i * 10, cigar=f"{n}M" if n else "*", seq=seq, qual=(None if (n and i % 17 == 0) else q) if n else None))
    data = bw.bam_bytes([("a", 1_000_000)], recs, text="@HD\tVN:1.6\n@SQ\tSN:a\tLN:1000000\n")
    exp = orc.bam_read(data)
    L = duckhts_amd.lib()
    L.dhts_bam_set_qual_packed.argtypes = [C.c_void_p, C.c_int]; L.dhts_bam_set_qual_packed.restype = None
    L.dhts_bam_batch_host_bytes.restype = C.c_uint64; L.dhts_bam_batch_host_bytes.argtypes = [C.c_void_p, C.c_uint32]
    L.dhts_host_alloc.restype = C.c_void_p; L.dhts_host_alloc.argtypes = [C.c_uint64]; L.dhts_host_free.argtypes = [C.c_void_p]
    L.dhts_bam_batch_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    L.dhts_bam_batch_fetch_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    L.dhts_bam_batch_fetch_wait.argtypes = [C.c_void_p, C.c_int]

    def decode(hb):
        n = hb.n_rows
        off = np.ctypeslib.as_array(C.cast(hb.qual.off, C.POINTER(C.c_uint32)), (n + 1,)); ln = np.ctypeslib.as_array(C.cast(hb.qual.len, C.POINTER(C.c_uint32)), (n,))
        raw = np.ctypeslib.as_array(C.cast(hb.qual.bytes, C.POINTER(C.c_uint8)), (int(hb.qual.nbytes),))
        if hb.qual_bits == 0:
            return [raw[off[i]:off[i] + ln[i]].tobytes() for i in range(n)]
        sym, stream, bts = raw[:16], raw[16:], hb.qual_bits
        out = []
        for i in range(n):
            ks = np.arange(int(off[i]), int(off[i]) + int(ln[i]), dtype=np.int64)
            codes = (stream[ks * bts // 8] >> ((ks * bts) % 8).astype(np.uint8)) & ((1 << bts) - 1)
            out.append(sym[codes].tobytes())
        return out

    for overlapped in (False, True):
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(data); ctx.bgzf_index(); ctx.bam_open()
            L.dhts_bam_set_qual_packed(ctx.h, 1)
            quals, arenas = [], []
            while True:
                b = ctx.next_batch(0)
                if b.n_rows:
                    assert b.qual_bits == 0                      # the batch in HBM is unchanged
                    need = L.dhts_bam_batch_host_bytes(C.byref(b), 0x1FFF)
                    arena = L.dhts_host_alloc(need); arenas.append(arena)
                    hb = duckhts_amd.BamBatch()
                    if overlapped:
                        assert L.dhts_bam_batch_fetch_begin(ctx.h, C.byref(b), 0x1FFF, arena, need, C.byref(hb), 0) == 0, ctx.L.dhts_error(ctx.h)
                        assert L.dhts_bam_batch_fetch_wait(ctx.h, 0) == 0
                    else:
                        assert L.dhts_bam_batch_fetch(ctx.h, C.byref(b), 0x1FFF, arena, need, C.byref(hb)) == 0, ctx.L.dhts_error(ctx.h)
                    assert hb.qual_bits == bits, (hb.qual_bits, bits)
                    quals += decode(hb)
                if b.status != 0:
                    break
            assert quals == list(exp["QUAL"])
            for a in arenas:
                L.dhts_host_free(a)
        finally:
            ctx.close()


# ---- next-batch prefetch: a caller that changes the batch size invalidates the prefetched phase B ---------------------------
@pytest.mark.gpu
def test_varying_batch_sizes_discard_the_prefetch(monkeypatch):
    monkeypatch.setenv("DHTS_PREFETCH", "1")        # the next batch's phase B beside the record stage is opt-in since round 3
    data = synth.bam_file(120000, seed=17)
    exp = orc.bam_read(data)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); ctx.bgzf_index(); hdr = ctx.bam_open()
        for sizes in ([3, 7, 2, 5, 11, 1], [4], [1, 64]):
            ctx.rewind()
            qn, pos, k = [], [], 0
            while True:
                b = ctx.next_batch(sizes[k % len(sizes)]); k += 1
                if b.n_rows:
                    h = ctx.batch_to_host(b, hdr)
                    qn += h["QNAME"]; pos += list(h["POS"])
                if b.status != 0:
                    assert b.status == 1
                    break
            assert qn == exp["QNAME"] and pos == list(exp["POS"])
    finally:
        ctx.close()


# ---- both LDS layouts of the Huffman kernel ---------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("nlo", ["196", "288"])
def test_phase_a_layouts_forced(nlo):
    """launches of > 98,304 blocks use the far-table layout (196 sorted symbols in LDS), smaller ones keep all 288 in LDS: force
    each on a file small enough for the oracle (the knob is read once per process, hence the child process)"""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import duckhts_amd, orc, cases\n"
        "from duckhts_amd import synth\n"
        "for data in (synth.bam_file(60000, seed=31), cases.ALL_CASES['fixed_huffman'](), cases.ALL_CASES['basic_stored'](), cases.ALL_CASES['basic_level9']()):\n"
        "    exp = orc.bam_read(data); got = duckhts_amd.read_bam(data, max_blocks=700)\n"
        "    assert got['n_rows'] == exp['n_rows'] and got['status'] == 1\n"
        "    for k in duckhts_amd.BAM_COLUMNS: assert list(got[k]) == list(exp[k]), k\n"
        "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, DHTS_PHASE_A_NLO=nlo)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


# ---- projection pushdown into the string pass ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cols", [("QNAME", "SEQ"), ("QUAL",), ("CIGAR", "READ_GROUP_ID"), ("FLAG", "POS")])
def test_projected_string_columns(cols):
    """dhts_bam_next_batch(colmask): heaps of unprojected string columns are not written; projected ones are unchanged"""
    data = synth.bam_file(30000, seed=3)
    exp = orc.bam_read(data)
    ids = {"QNAME": 0, "FLAG": 1, "POS": 3, "CIGAR": 5, "SEQ": 9, "QUAL": 10, "READ_GROUP_ID": 11}
    mask = 0
    for c in cols:
        mask |= 1 << ids[c]
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); ctx.bgzf_index(); ctx.bam_open()
        row = 0
        while True:
            b = ctx.next_batch(5, colmask=mask)
            n = int(b.n_rows)
            for c in cols:
                if c in ("FLAG", "POS"):
                    got = ctx.d2h(b.flag if c == "FLAG" else b.pos, n, np.uint16 if c == "FLAG" else np.int64)
                    assert got.tolist() == list(exp[c][row:row + n])
                    continue
                col = {"QNAME": b.qname, "CIGAR": b.cigar, "SEQ": b.seq, "QUAL": b.qual, "READ_GROUP_ID": b.rg}[c]
                off = ctx.d2h(col.off, n + 1, np.uint32); ln = ctx.d2h(col.len, n, np.uint32)
                heap = ctx.d2h(col.bytes, int(col.nbytes), np.uint8).tobytes()
                for i in range(n):
                    e = exp[c][row + i]
                    if e is not None:
                        assert heap[off[i]:off[i] + ln[i]] == bytes(e), (c, row + i)
            row += n
            if b.status != 0:
                break
        assert row == exp["n_rows"]
    finally:
        ctx.close()


# ---- interval overlap join ----------------------------------------------------------------
def _overlap_check(data, tid, beg, end, region=None, max_blocks=0):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import region_oracle as ro
    t = orc.bam_read(data)
    exp = ro.overlap_join(t, tid, beg, end)
    if region is not None:
        keep = ro.keep_mask(t, region)
        exp = [e for e, k in zip(exp, keep) if k]
    got = duckhts_amd.read_bam(data, overlap=(tid, beg, end), region=region, max_blocks=max_blocks)
    assert got["n_rows"] == len(exp)
    assert len(got["OVERLAPS"]) == len(exp)
    for i, (a, b) in enumerate(zip(got["OVERLAPS"], exp)):
        assert np.sort(a).tolist() == b.tolist(), (i, a, b)
    return sum(len(x) for x in exp)


@pytest.mark.gpu
def test_overlap_join_golden_cgranges_vector():
    """the pairs the reference's own cgranges reports for 400 intervals x the reads of range.bam (tests/golden/overlap_range_bam.json)"""
    import json
    g = json.loads(read_golden("overlap_range_bam.json"))
    data = read_golden("range.bam")
    got = duckhts_amd.read_bam(data, overlap=(g["tid"], g["beg"], g["end"]))
    assert got["n_rows"] == 112 and sum(len(x) for x in got["OVERLAPS"]) == 866
    for a, b in zip(got["OVERLAPS"], g["overlaps"]):
        assert np.sort(a).tolist() == b
    # intervals on contigs the header does not have never match; an empty set switches the join off
    got = duckhts_amd.read_bam(data, overlap=([99, -1], [0, 0], [10 ** 9, 10 ** 9]))
    assert sum(len(x) for x in got["OVERLAPS"]) == 0
    assert "OVERLAPS" in duckhts_amd.read_bam(data, overlap=([], [], []))


@pytest.mark.gpu
def test_overlap_join_synthetic_batches_and_regions():
    data = synth.bam_file(60000, seed=9)
    hdr = duckhts_amd.read_bam(data, max_blocks=0)["header"]
    n_ref = len(hdr["ref_names"])
    rng = np.random.default_rng(21)
    n = 20000
    tid = rng.integers(0, n_ref, n).astype(np.int32)
    beg = rng.integers(0, 60_000_000, n).astype(np.int64)
    end = beg + rng.choice([0, 1, 100, 1000, 50_000, 5_000_000], n)      # nested long intervals exercise the running-max bound
    assert _overlap_check(data, tid, beg, end, max_blocks=5) > 0
    assert _overlap_check(data, tid, beg, end, region="chr1:1-30,000,000,chr2", max_blocks=3) > 0
    # duplicates and identical starts keep their own ids
    tid2 = np.zeros(6, np.int32); beg2 = np.array([0, 0, 0, 10, 10, 10], np.int64); end2 = np.array([10 ** 9] * 6, np.int64)
    _overlap_check(data, tid2, beg2, end2)


def _bed_join_check(data, bed, max_blocks=0, as_path=None):
    """read_bam x BED: the device's read_bed rows and join against the oracle's (bed_rows -> overlap_join)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import region_oracle as ro
    t = orc.bam_read(data)
    rows = ro.bed_rows(bed)
    exp = ro.overlap_join(t, *ro.bed_join_intervals(rows, [bytes(x) for x in t["ref_names"]]))
    got = duckhts_amd.read_bam(data, overlap_bed=as_path if as_path is not None else bed, max_blocks=max_blocks)
    assert got["n_bed_rows"] == len(rows)
    assert got["n_rows"] == len(exp) == len(got["OVERLAPS"])
    for i, (a, b) in enumerate(zip(got["OVERLAPS"], exp)):
        assert np.sort(a).tolist() == b.tolist(), (i, a, b)
    return rows, exp, t


@pytest.mark.gpu
def test_overlap_join_with_the_reference_bed(tmp_path):
    """the reference's test/data/targets.bed x range.bam: read_bed's rows (duckhts.test:241-251) as the join's intervals, checked against the
    oracle and, where it is built, against the reference's own cgranges"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import region_oracle as ro
    data, bed = read_golden("range.bam"), read_golden("targets.bed")
    rows, exp, t = _bed_join_check(data, bed)
    assert len(rows) == 4                                  # (the targets lie in the first 20 bases; the reads of range.bam begin near 900: no pairs)
    # the same targets moved onto the reads: 1 kb further along, each 100 times as long
    moved = b"".join(b"\t".join([f[0], b"%d" % (int(f[1]) * 100 + 1000), b"%d" % (int(f[2]) * 100 + 1000)] + f[3:]) + b"\n" for f in (ln.split(b"\t") for ln in bed.splitlines() if ln))
    rows, exp, t = _bed_join_check(data, moved)
    assert len(rows) == 4 and sum(len(x) for x in exp) > 20
    lib = os.path.join(ROOT, "oracle", "_ref", "libcgranges.so")
    if os.path.exists(lib):
        names = [bytes(x).decode() for x in t["ref_names"]]
        tid, beg, end = ro.bed_join_intervals(rows, t["ref_names"])
        q = []
        for i in range(t["n_rows"]):
            st = int(t["POS"][i]) - 1
            q.append((names[int(t["tid"][i])], st, ro.endpos(st, int(t["FLAG"][i]), t["CIGAR"][i])))
        ref = ro.cgranges_overlap(lib, names, tid, beg, end, q)
        assert all(a.tolist() == b.tolist() for a, b in zip(exp, ref))
    # the same BED as a file: plain, and bgzipped by the device (duckhts.test:260-284 read the .bed.gz)
    p = tmp_path / "targets.bed"; p.write_bytes(bed)
    _bed_join_check(data, bed, as_path=str(p))
    pz = tmp_path / "targets.bed.gz"; 
    cx = duckhts_amd.Context(0)
    try:
        pz.write_bytes(cx.bgzf_compress(bed))
    finally:
        cx.close()
    _bed_join_check(data, bed, as_path=str(pz))


@pytest.mark.gpu
def test_overlap_join_bed_line_rules_and_size():
    data = synth.bam_file(30000, seed=4)
    hdr = duckhts_amd.read_bam(data)["header"]
    names = [bytes(x) for x in hdr["ref_names"]]
    rng = np.random.default_rng(8)
    lines = [b"#header", b"track name=t", b"browser position chr1:1-2", b""]
    for k in range(50000):                              # sorted runs of the same chrom, a few unknown names, NULL fields, CRLF, extra columns
        nm = names[int(rng.integers(0, len(names)))] if k % 97 else b"chrUn_%d" % k
        b = int(rng.integers(0, 50_000_000)); e = b + int(rng.choice([0, 1, 150, 4000, 2_000_000]))
        f1 = b"%d" % b if k % 211 else b"12x"
        f2 = b"%d" % e if k % 307 else b""
        ln = nm + b"\t" + f1 + b"\t" + f2 + (b"\tname%d\t0\t+" % k if k % 3 == 0 else b"") + (b"\r" if k % 5 == 0 else b"")
        lines.append(ln)
        if k % 1000 == 0:
            lines.append(b"")
    bed = b"\n".join(lines)                            # no newline at the end of the last line
    rows, exp, _ = _bed_join_check(data, bed, max_blocks=4)
    assert len(rows) == 50000 and sum(len(x) for x in exp) > 1000
    _bed_join_check(data, bed + b"\n")
    # a line with fewer than 3 fields is read_bed's error; empty text = no intervals
    with pytest.raises(duckhts_amd.DhtsError, match="fewer than 3"):
        duckhts_amd.read_bam(data, overlap_bed=b"chr1\t5\t9\nchr1\t7\n")
    got = duckhts_amd.read_bam(data, overlap_bed=b"#only a comment\n")
    assert got["n_bed_rows"] == 0 and sum(len(x) for x in got["OVERLAPS"]) == 0


# ---- standard_tags (row A5) ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("which", ["sam_equiv", "matrix", "fuzz", "fuzz_small_batches", "golden_range"])
def test_std_tag_columns(which):
    import tag_cases
    data = {"sam_equiv": tag_cases.aux_tags_sam_equivalent, "matrix": tag_cases.type_matrix, "fuzz": tag_cases.fuzz, "fuzz_small_batches": tag_cases.fuzz,
            "golden_range": lambda: read_golden("range.bam")}[which]()
    exp = orc.bam_read_std_tags(data)
    got = duckhts_amd.read_bam(data, std_tags_cols=list(range(56)), max_blocks=3 if which == "fuzz_small_batches" else 0)
    assert got["n_rows"] == exp["n_rows"]
    d = orc.bcf_cols_diff(exp, got["tags"])
    assert d is None, d
    # a projection of a few tag columns, in another order
    sel = [34, 48, 29, 7, 53]
    got = duckhts_amd.read_bam(data, std_tags_cols=sel)
    sub = {"n_rows": exp["n_rows"], "cols": [exp["cols"][i] for i in sel]}
    d = orc.bcf_cols_diff(sub, got["tags"])
    assert d is None, d


@pytest.mark.gpu
@pytest.mark.parametrize("excl", [True, False])
@pytest.mark.parametrize("which", ["sam_equiv", "matrix", "fuzz", "golden_range"])
def test_auxiliary_tags_map(which, excl):
    """AUXILIARY_TAGS (duckhts.test:179-185: map_extract(AUXILIARY_TAGS,'XZ') = [foo]) against the oracle"""
    import tag_cases
    data = {"sam_equiv": tag_cases.aux_tags_sam_equivalent, "matrix": tag_cases.type_matrix, "fuzz": tag_cases.fuzz, "golden_range": lambda: read_golden("range.bam")}[which]()
    exp = orc.bam_read_aux_map(data, excl)
    got = duckhts_amd.read_bam(data, aux_map="exclude_standard" if excl else "all", max_blocks=5)
    d = orc.bcf_cols_diff({"n_rows": exp["n_rows"], "cols": exp["cols"]}, got["aux"])
    assert d is None, d
    if which == "sam_equiv" and excl:
        k, v = (orc.bcf_col_py(c) for c in got["aux"]["cols"])
        assert k == [[b"XZ"]] and v == [[b"foo"]]


# ---- the two phase-A kernels produce the same scratch, word for word ---------------------------------------------------------
def _huff_scratch(ctx, nb, kernel):
    import ctypes as C
    L = ctx.L
    L.dhts_debug_huff_run.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
    L.dhts_debug_scratch_get.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.dhts_debug_huff_run(C.c_void_p(ctx.h), 0, nb, kernel) == 0, L.dhts_error(ctx.h)
    out = []
    meta = (C.c_uint32 * 4)(); lit = (C.c_uint8 * 65536)(); tok = (C.c_uint32 * 22528)()
    for s in range(nb):
        assert L.dhts_debug_scratch_get(C.c_void_p(ctx.h), s, meta, lit, tok) == 0, L.dhts_error(ctx.h)
        m = list(meta)
        if m[3] != 0:
            out.append(("failed",))
        else:
            out.append((m[0], m[1], m[2], bytes(bytearray(lit)[:m[1]]), bytes(bytearray(tok)[:4 * m[0]])))
    return out


def _replay_hits_bad_distance(blk):
    ntok, nlit, outlen, lit, tok = blk
    import struct
    pos = 0
    for (t,) in struct.iter_unpack("<I", tok):
        run = t >> 23
        pos += run
        if run != 511:
            if (t & 0x7fff) + 1 > pos:
                return True
            pos += ((t >> 15) & 255) + 3
    return False


@pytest.mark.gpu
def test_wave_and_lane_huffman_kernels_agree():
    """bgzf_huff_decode_wave (one wave per block, lookup tables, self-synchronising bit ranges; into fixed slots (2) and into the packed
    pool of the product path (3)) against bgzf_huff_decode (one lane per block, canonical arithmetic), both LDS layouts: literal stream,
    token stream and meta of every block, incl. damaged blocks"""
    import random
    files = [read_golden("range.bam"), read_golden("vcf_file.bcf"), read_golden("bgzf_boundaries3.bam"), synth.bam_file(40000, seed=5)]
    files += [cases.ALL_CASES[k]() for k in ("fixed_huffman", "basic_stored", "basic_level1", "basic_level9", "basic_tiny_blocks", "bad_deflate", "long_record")]
    import bamwriter as bw
    for name, raw in sorted(_lz_payloads().items()):
        files.append(bw.bgzf_file(raw, payload=65280, level=6))
    rnd = random.Random(9)
    dam = bytearray(synth.bam_file(20000, seed=6))
    for _ in range(60):                                    # payload damage: both kernels must fail the same blocks
        i = rnd.randrange(2000, len(dam) - 2000)
        dam[i] ^= 1 << rnd.randrange(8)
    files.append(bytes(dam))
    for data in files:
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(data)
            try:
                nb = ctx.bgzf_index()
            except duckhts_amd.DhtsError:
                continue
            if nb <= 0:
                continue
            ref = _huff_scratch(ctx, nb, 0)
            for kernel in (1, 2, 3):
                got = _huff_scratch(ctx, nb, kernel)
                for b, (x, y) in enumerate(zip(ref, got)):
                    if kernel >= 2 and x == ("failed",) and y != ("failed",):
                        # the wave kernel leaves "distance reaches in front of the block" to bgzf_lz_resolve: its tokens must trip that test
                        assert _replay_hits_bad_distance(y), f"block {b}: the lane kernel rejects it, the wave kernel's tokens replay"
                        continue
                    assert x == y, f"block {b} differs between kernel 0 and kernel {kernel}"
        finally:
            ctx.close()


# ---- region queries through several disjoint index windows ---------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("gap_mb", ["0", "0.05", "32"])
def test_region_query_disjoint_index_windows(gap_mb):
    """the chunk list of a multi-region query (hts_itr_multi_bam / reg2intervals, hts.c:3597-3739, 3299-3354) as disjoint scan windows:
    DHTS_WINDOW_GAP_MB=0 keeps every merged chunk a window of its own, 32 (the default) merges everything in a file this small.
    The rows must be those of the unindexed predicate -- and of the region oracle -- whatever the windows, in file order, no duplicates."""
    import region_oracle
    data = synth.bam_file(300000, seed=29)
    exp = orc.bam_read(data)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); ctx.bgzf_index(); ctx.bam_open()
        bai = ctx.build_index()
    finally:
        ctx.close()
    regions = ["chr1:1,000,000-1,200,000,chr5:40000000-41000000,chrX:1-3,000,000,chr1:200,000,000-201,000,000",
               "chr2:5000000-5100000,chr2:5,050,000-5,300,000,chr20,chr3:100-200",
               ",".join(f"chr{1 + k % 22}:{1_000_000 * (1 + 7 * k % 40)}-{1_000_000 * (1 + 7 * k % 40) + 150_000}" for k in range(60))]
    old = os.environ.get("DHTS_WINDOW_GAP_MB")
    os.environ["DHTS_WINDOW_GAP_MB"] = gap_mb
    try:
        for region in regions:
            keep = region_oracle.keep_mask(exp, region)
            want = [q for q, k in zip(exp["QNAME"], keep) if k]
            for mb in (0, 5):
                a = duckhts_amd.read_bam(data, region=region, index=bai, max_blocks=mb)
                assert a["status"] == 1 and a["n_rows"] == len(want) and a["QNAME"] == want, (region[:40], gap_mb, mb, a["n_rows"], len(want))
                assert list(a["POS"]) == [int(p) for p, k in zip(exp["POS"], keep) if k]
    finally:
        if old is None:
            del os.environ["DHTS_WINDOW_GAP_MB"]
        else:
            os.environ["DHTS_WINDOW_GAP_MB"] = old


def test_region_query_stages_only_its_index_windows(tmp_path):
    """dhts_bam_region_segments + dhts_open_path_segments: header blocks + the index windows are the only resident bytes; rows, order and
    the virtual offsets of the hand-off are those of the whole-file scan (and of the region oracle)."""
    import ctypes as C
    import region_oracle
    data = synth.bam_file(300000, seed=31)
    path = os.path.join(str(tmp_path), "w.bam")
    open(path, "wb").write(data)
    exp = orc.bam_read(data)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); ctx.bgzf_index(); ctx.bam_open()
        bai = ctx.build_index()
        L = duckhts_amd.lib()
        L.dhts_bam_header_bytes.restype = C.c_uint64
        L.dhts_bam_header_bytes.argtypes = [C.c_void_p]
        hb = L.dhts_bam_header_bytes(ctx.h)
        regions = ["chr1:1,000,000-1,200,000", "chr1:1,000,000-1,200,000,chr5:40000000-41000000,chrX:1-3,000,000,chr1:200,000,000-201,000,000",
                   "chr2:5000000-5100000,chr2:5,050,000-5,300,000,chr20,chr3:100-200", "chrM", "chr22:50,000,000-60,000,000,*", "*", "chrY:1-10",
                   ",".join(f"chr{1 + k % 22}:{1_000_000 * (1 + 7 * k % 40)}-{1_000_000 * (1 + 7 * k % 40) + 150_000}" for k in range(60))]
        old = os.environ.get("DHTS_WINDOW_GAP_MB")
        try:
            for gap in ("0", "0.25", "32"):
                os.environ["DHTS_WINDOW_GAP_MB"] = gap
                for region in regions:
                    assert ctx.set_regions(region)
                    seg = ctx.region_segments(bai)
                    assert seg is not None
                    keep = region_oracle.keep_mask(exp, region)
                    want = [q for q, k in zip(exp["QNAME"], keep) if k]
                    for mb in (0, 3):
                        a = duckhts_amd.read_bam(path, region=region, index=bai, max_blocks=mb, sparse=(hb, seg[0], seg[1]))
                        assert a["status"] == 1 and a["n_rows"] == len(want) and a["QNAME"] == want, (region[:40], gap, mb, a["n_rows"], len(want))
                        assert list(a["POS"]) == [int(p) for p, k in zip(exp["POS"], keep) if k]
                    if gap == "0" and region == regions[0]:
                        staged = hb + sum(int(e) - int(b) for b, e in zip(*seg)) + 65536 * len(seg[0])
                        assert staged < len(data) // 10, (staged, len(data))             # a 200 kb region of a 3.1 Gb genome: a sliver of the file
            assert ctx.set_regions("chr1")                                               # a whole-file query has no segments
        finally:
            if old is None:
                del os.environ["DHTS_WINDOW_GAP_MB"]
            else:
                os.environ["DHTS_WINDOW_GAP_MB"] = old
    finally:
        ctx.close()


def test_whole_file_stays_resident_for_the_next_query(tmp_path):
    """a file staged whole is kept in HBM (tagged pool buffer) when its context goes away; the next context on the unchanged file takes it
    over without reading the file; a changed file (size / mtime) is staged afresh"""
    import ctypes as C
    L = duckhts_amd.lib()
    L.dhts_resident_from_cache.argtypes = [C.c_void_p]
    data = synth.bam_file(60000, seed=5)
    path = os.path.join(str(tmp_path), "c.bam")
    open(path, "wb").write(data)
    exp = orc.bam_read(data)

    def scan(asynchronous):
        ctx = duckhts_amd.Context(0)
        try:
            if asynchronous:
                L.dhts_open_path_async.argtypes = [C.c_void_p, C.c_char_p]
                L.dhts_stage_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int)]
                L.dhts_stage_wait.restype = C.c_int64
                L.dhts_bgzf_index_staged.argtypes = [C.c_void_p]
                L.dhts_bgzf_index_staged.restype = C.c_int64
                assert L.dhts_open_path_async(ctx.h, os.fsencode(path)) == 0
                done = C.c_int(0)
                assert L.dhts_stage_wait(ctx.h, 1 << 62, C.byref(done)) == os.path.getsize(path) and done.value == 1
                assert L.dhts_bgzf_index_staged(ctx.h) > 0
            else:
                ctx.open(path)
                ctx.bgzf_index()
            hit = L.dhts_resident_from_cache(ctx.h)
            hdr = ctx.bam_open()
            names = []
            while True:
                b = ctx.next_batch(0)
                if b.n_rows:
                    names += ctx.batch_to_host(b, hdr)["QNAME"]
                if b.status != 0:
                    break
            return hit, names
        finally:
            ctx.close()

    L.dhts_release_pools()
    hit, names = scan(False)
    assert hit == 0 and names == exp["QNAME"]
    for asynchronous in (False, True, True, False):
        hit, names = scan(asynchronous)
        assert hit == 1 and names == exp["QNAME"]
    data2 = synth.bam_file(50000, seed=6)
    open(path, "wb").write(data2)
    hit, names = scan(False)
    assert hit == 0 and names == orc.bam_read(data2)["QNAME"]
    hit, names = scan(True)
    assert hit == 1
    L.dhts_release_pools()
    hit, names = scan(True)                                     # staged by the background readers: tagged when the context goes away
    assert hit == 0 and names == orc.bam_read(data2)["QNAME"]
    hit, names = scan(False)
    assert hit == 1
    L.dhts_release_pools()


def test_fused_inflate_option_in_a_child_process(tmp_path):
    """DHTS_INFLATE=fused (one wave decodes AND resolves a block: bgzf_inflate_fused) is read once per process: a child scans a
    multi-batch file and the golden files with it, column digests against the oracle"""
    import subprocess
    import sys
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import duckhts_amd, orc
from duckhts_amd import synth
for d in (synth.bam_file(120000, seed=23), open(os.path.join(%r, "tests", "golden", "range.bam"), "rb").read(), synth.bam_file(3000, seed=4, payload=700)):
    exp = orc.bam_read(d)
    for mb in (0, 7):
        got = duckhts_amd.read_bam(d, max_blocks=mb)
        assert got["n_rows"] == exp["n_rows"], (got["n_rows"], exp["n_rows"])
        for k in duckhts_amd.BAM_COLUMNS:
            assert list(got[k]) == list(exp[k]), k
print("fused ok")
""" % (ROOT, ROOT, ROOT)
    env = dict(os.environ, DHTS_INFLATE="fused")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "fused ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_fused_row_pass_option_in_a_child_process(tmp_path):
    """DHTS_ROWS=fused (bam_tile_rows: unpack + heap offsets by decoupled look-back + strings on one staging of every tile, one host
    round trip per batch) is read once per process: a child scans multi-batch files, the golden file, long records, a projection without
    strings and a damaged record with it -- every column against the oracle"""
    import subprocess
    import sys
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import duckhts_amd, orc
from duckhts_amd import synth
import cases
files = [synth.bam_file(120000, seed=23), open(os.path.join(%r, "tests", "golden", "range.bam"), "rb").read(), synth.bam_file(3000, seed=4, payload=700)]
files += [cases.ALL_CASES[k]() for k in sorted(cases.ALL_CASES)]
for d in files:
    exp = orc.bam_read(d)
    for mb in (0, 7, 2):
        got = duckhts_amd.read_bam(d, max_blocks=mb)
        assert got["n_rows"] == exp["n_rows"], (got["n_rows"], exp["n_rows"])
        for k in duckhts_amd.BAM_COLUMNS:
            assert list(got[k]) == list(exp[k]), k
print("fused rows ok")
""" % (ROOT, ROOT, ROOT)
    env = dict(os.environ, DHTS_ROWS="fused")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "fused rows ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_block_table_extended_piece_by_piece_equals_the_one_shot_table():
    """a file that is still being staged: the block table is extended behind its last complete block (index_extend_tail) instead of being
    rebuilt over the whole prefix -- the table at the end, and the scan that runs on it while it grows, equal those of the whole file"""
    import ctypes as C
    L = duckhts_amd.lib()
    L.dhts_debug_index_prefix.argtypes = [C.c_void_p, C.c_uint64]
    L.dhts_debug_index_prefix.restype = C.c_int64
    L.dhts_blocks_ahead.argtypes = [C.c_void_p]
    L.dhts_blocks_ahead.restype = C.c_int64
    for payload, step in ((20000, 1 << 20), (65280, 3 << 20), (777, 100000)):
        data = synth.bam_file(150000, seed=11, payload=payload)
        exp = orc.bam_read(data)
        ref = duckhts_amd.Context(0)
        try:
            ref.open(data); nb_ref = ref.bgzf_index(); table_ref = ref.bgzf_table(nb_ref)
        finally:
            ref.close()
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(data)
            hdr = None; names = []; pos = []; extensions = 0; finished = False; upto = step; last_nb = 0
            while not finished:
                nb = L.dhts_debug_index_prefix(ctx.h, upto); extensions += 1
                assert nb >= last_nb, duckhts_amd.last_error(ctx) if hasattr(duckhts_amd, "last_error") else nb
                last_nb = nb
                whole = upto >= len(data)
                if nb > 0 and hdr is None:
                    hdr = ctx.bam_open()
                while hdr is not None:
                    if not whole and L.dhts_blocks_ahead(ctx.h) < 8:
                        break                                   # leave the last blocks: a record may run into bytes that "are not there yet"
                    b = ctx.next_batch(5)
                    if b.n_rows:
                        h = ctx.batch_to_host(b, hdr)
                        names += h["QNAME"]; pos += list(h["POS"])
                    if b.status != 0:
                        assert b.status == 1
                        finished = True
                        break
                upto += step
            assert extensions >= 4
            assert names == exp["QNAME"] and pos == list(exp["POS"]), (payload, len(names), len(exp["QNAME"]))
            table = ctx.bgzf_table(nb_ref)
            assert table[3] == table_ref[3]
            for a_, b_ in zip(table[:3], table_ref[:3]):
                assert (a_ == b_).all(), payload
        finally:
            ctx.close()


@pytest.mark.gpu
def test_concurrent_contexts_in_one_process():
    """an engine runs several table functions at once (a join of two read_bam calls, a UNION of files): contexts are independent, the device /
    pinned / stream pools are shared -- four threads scanning BAM, BCF, VCF text and compressing at the same time get what they get alone"""
    import threading
    import gzip
    import vep_cases
    bam = cases.case_basic(payload=4000, n=20000, seed=31)
    bcf = read_golden("vcf_file.bcf")
    txt = vep_cases.fixture_text().encode()
    want_bam, want_bcf, want_txt = orc.bam_read(bam), orc.bcf_read(bcf), orc.bcf_read(txt)
    errs = []

    def run_bam():
        for _ in range(4):
            got = duckhts_amd.read_bam(bam, max_blocks=7)
            if got["n_rows"] != want_bam["n_rows"] or list(got["POS"]) != list(want_bam["POS"]) or got["SEQ"] != list(want_bam["SEQ"]):
                errs.append("bam")

    def run_bcf():
        for _ in range(6):
            if orc.bcf_cols_diff(want_bcf, duckhts_amd.read_bcf(bcf)) is not None:
                errs.append("bcf")

    def run_txt():
        for _ in range(4):
            if orc.bcf_cols_diff(want_txt, duckhts_amd.read_bcf(txt, max_blocks=2)) is not None:
                errs.append("txt")

    def run_zip():
        ctx = duckhts_amd.Context(0)
        try:
            for _ in range(4):
                if gzip.decompress(ctx.bgzf_compress(txt)) != txt:
                    errs.append("bgzip")
        finally:
            ctx.close()

    th = [threading.Thread(target=f) for f in (run_bam, run_bcf, run_txt, run_zip, run_bam)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
