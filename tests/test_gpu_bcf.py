"""GPU parity tests for read_bcf: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import os

import numpy as np
import pytest

import bcf_cases
import orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold(name):
    with open(os.path.join(GOLD, name), "rb") as f:
        return f.read()


def _check(data, tidy=False, **kw):
    import duckhts_amd
    exp = orc.bcf_read(data, tidy)
    got = duckhts_amd.read_bcf(data, tidy=tidy, **kw)
    d = orc.bcf_cols_diff(exp, got)
    assert d is None, d
    assert (got["status"] == 1) == (exp["status"] == 0), (got["status"], exp["status"])
    return exp, got


def test_golden_vcf_file_bcf():
    exp, got = _check(_gold("vcf_file.bcf"))
    assert got["n_rows"] == 15 and len(got["cols"]) == 23
    col = {c["name"]: orc.bcf_col_py(c) for c in got["cols"]}
    assert (col["CHROM"][0], col["POS"][0], col["FORMAT_GT_A"][0], col["FORMAT_GQ_A"][0]) == (b"1", 3000150, b"0/1", 245)   # duckhts.test:28-62
    assert col["INFO_TEST"][3] == 5 and col["FILTER"][2] == [b"q10"]


def test_golden_tidy():
    exp, got = _check(_gold("vcf_file.bcf"), tidy=True)
    assert got["n_rows"] == 30


@pytest.mark.parametrize("name,data,tidy", bcf_cases.all_cases(), ids=[c[0] for c in bcf_cases.all_cases()])
def test_cases(name, data, tidy):
    _check(data, tidy)


@pytest.mark.parametrize("mb", [1, 2, 3])
def test_small_batches_carry(mb):
    cases = {n: (d, t) for n, d, t in bcf_cases.all_cases()}
    for name in ("fuzz_small_blocks", "fuzz_tidy", "long_record", "bad_info_key", "truncated_mid_record", "basic"):
        d, t = cases[name]
        _check(d, t, max_blocks=mb)


def test_header_errors():
    import duckhts_amd
    for name, data in bcf_cases.header_error_cases():
        with pytest.raises(duckhts_amd.DhtsError) as e:
            duckhts_amd.read_bcf(data)
        if name != "not_bgzf":
            assert "Failed to read BCF/VCF header" in str(e.value), (name, str(e.value))


def test_projection_subsets():
    import duckhts_amd
    data = dict((n, d) for n, d, _ in bcf_cases.all_cases())["fuzz_small_blocks"]
    exp = orc.bcf_read(data)
    for proj in (["CHROM"], ["POS", "QUAL"], ["FORMAT_GT_S2", "INFO_TAGS", "ALT"], ["FILTER", "ID", "FORMAT_PL_S3", "INFO_DB", "REF"]):
        got = duckhts_amd.read_bcf(data, columns=proj)
        sub = {"n_rows": exp["n_rows"], "cols": [exp["by_name"][p] for p in proj]}
        d = orc.bcf_cols_diff(sub, got)
        assert d is None, (proj, d)


def test_sharded_block_ranges_concatenate():
    """BGZF block-range shards (speculative first record, halo for the last) reproduce the sequential scan."""
    import duckhts_amd
    data = dict((n, d) for n, d, _ in bcf_cases.all_cases())["fuzz_small_blocks"]
    exp = orc.bcf_read(data)
    ctx = duckhts_amd.Context(0)
    ctx.open(data)
    nb = ctx.bgzf_index()
    ctx.close()
    cuts = [0, nb // 3, (2 * nb) // 3, nb]
    spans, rows = [], 0
    parts = []
    for r in range(3):
        got = duckhts_amd.read_bcf(data, block_range=(cuts[r], cuts[r + 1], r > 0))
        spans.append((got["first_rec_uoff"], got["end_uoff"], got["n_rows"]))
        parts.append(got)
    assert duckhts_amd.check_handoff(spans) == exp["n_rows"]
    pos = np.concatenate([p["by_name"]["POS"]["fixed"] for p in parts])
    assert np.array_equal(pos, exp["by_name"]["POS"]["fixed"])
    gq = np.concatenate([p["by_name"]["FORMAT_GQ_S1"]["fixed"] for p in parts])
    assert np.array_equal(gq, exp["by_name"]["FORMAT_GQ_S1"]["fixed"])


@pytest.mark.parametrize("tidy", [False, True])
def test_synthetic_config3_shape(tidy):
    """SURVEY.md 8(d) config 3 shape: 16 samples, typed INFO/FORMAT, int8/16/32 widths, missing values; multi-batch."""
    from duckhts_amd import synth
    data = synth.bcf_file(30000 if tidy else 60000, seed=43)
    exp, got = _check(data, tidy, max_blocks=64)
    assert len(got["cols"]) == (7 + 8 + 1 + 6 if tidy else 7 + 8 + 96)
    assert got["n_rows"] == (30000 * 16 if tidy else 60000)


def _region_check(data, region, tidy=False, **kw):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "oracle"))
    import duckhts_amd
    import region_oracle
    exp = orc.bcf_read(data, tidy)
    ctx = duckhts_amd.Context(0)
    ctx.open(data); ctx.bgzf_index()
    contigs = [c.decode() if c else "\x01" for c in duckhts_amd.BcfScan(ctx, tidy).contigs]
    ctx.close()
    rows = region_oracle.bcf_region_rows(exp, contigs, region, exp["n_samples"] if tidy and exp["n_samples"] else 1)
    want = orc.bcf_take_rows(exp, rows)
    got = duckhts_amd.read_bcf(data, tidy=tidy, region=region, **kw)
    d = orc.bcf_cols_diff(want, got)
    assert d is None, (region, d)
    return got["n_rows"]


def test_region_golden_counts():
    """duckhts.test:88-105: single region 2 rows; multi-region = chained union"""
    data = _gold("vcf_file.bcf")
    assert _region_check(data, "1:3000150-3000151") == 2
    assert _region_check(data, "1:3062915-3062915") == 2
    assert _region_check(data, "1:3000150-3000151,1:3062915-3062915") == 4
    assert _region_check(data, "1:3000150-3000151,1:3000150-3000151") == 4          # overlapping regions repeat rows (bcf_reader.c:930-932)
    assert _region_check(data, "nosuch,2,.") == 1 + 15
    assert _region_check(data, "4:3,258,448", tidy=True) >= 0
    assert _region_check(data, "nosuch") == 0


def test_region_with_csi_window():
    """the CSI index (BGZF-compressed, inflated on the device) narrows the scan window; rows are unchanged"""
    import duckhts_amd
    data, csi = _gold("vcf_file.bcf"), _gold("vcf_file.bcf.csi")
    for region, want in (("1:3000150-3000151", 2), ("1:3062915-3062915", 2), ("4", 2), ("2:1-10", 0), ("1:3000150-3000151,4:3258448-3258448", 3)):
        a = duckhts_amd.read_bcf(data, region=region)
        b = duckhts_amd.read_bcf(data, region=region, index=csi)
        assert a["n_rows"] == b["n_rows"] == want, (region, a["n_rows"], b["n_rows"])
        assert orc.bcf_cols_diff(a, b) is None
    # the window really is narrower than the file for a late contig (first record offset moves past the header end)
    ctx = duckhts_amd.Context(0)
    ctx.open(data); ctx.bgzf_index()
    sc = duckhts_amd.BcfScan(ctx)
    assert sc.set_region("4")
    sc.load_index(csi)
    b = sc.next_batch()
    assert b.n_rows == 2 and b.first_rec_uoff > sc.first_rec_uoff
    ctx.close()


def test_region_synthetic():
    from duckhts_amd import synth
    data = synth.bcf_file(40000, seed=9)
    for region in ("chr1:1,000,000-30,000,000", "chr2,chrX:1-50000000", "chr21:1-1000"):
        _region_check(data, region, max_blocks=16)


# ---- a context returns every byte of HBM when it is destroyed (DevBuf is RAII now) -------------------------
@pytest.mark.gpu
def test_no_hbm_leak_over_200_queries():
    import ctypes as C
    import duckhts_amd
    import tag_cases
    data = _gold("vcf_file.bcf")
    bam = _gold("range.bam")
    tags = tag_cases.fuzz(seed=3, n=200, payload=3000)

    def cycle(i):
        duckhts_amd.read_bcf(data, tidy=bool(i & 1))
        if i % 4 == 0:
            duckhts_amd.read_bam(bam, region="CHROMOSOME_I:1-5000")
            duckhts_amd.read_bam(tags, std_tags_cols=list(range(56)), aux_map="all")

    L = duckhts_amd.lib()

    def free_hbm():
        f, t = C.c_uint64(), C.c_uint64()
        assert L.dhts_device_mem_info(0, C.byref(f), C.byref(t)) == 0
        return f.value

    for i in range(8):                      # warm up: code objects, runtime pools
        cycle(i)
    L.dhts_release_pools()                  # buffers of destroyed contexts are pooled for the next query: hand the idle ones back
    free0 = free_hbm()
    held = []
    for i in range(200):
        cycle(i)
        if i % 50 == 49:
            held.append(free0 - free_hbm())
    assert max(held) - min(held) < (64 << 20), f"the pool keeps growing: {held}"        # steady state, not a leak with a pool in front
    L.dhts_release_pools()
    free1 = free_hbm()
    # What the HIP runtime keeps for itself (scratch and kernel-argument room of the queues behind the pooled streams) is sized on first use and
    # depends on which queues the tests before this one have touched; a leak grows with every cycle.  So: a second run of cycles may not cost more.
    assert free0 - free1 < (64 << 20), f"{(free0 - free1) >> 10} KiB of HBM lost over 200 create -> scan -> destroy cycles"
    for i in range(200):
        cycle(i)
    L.dhts_release_pools()
    free2 = free_hbm()
    assert free1 - free2 < (4 << 20), f"{(free1 - free2) >> 10} KiB of HBM lost over 200 more cycles"


def test_format_flag_column_stays_inside_its_payload():
    """a FORMAT Flag binds as a BOOLEAN column that is always NULL; clearing its one-byte payload used a 4-byte store and ran into the
    next column's values (found by the VCF text cases of round 2)"""
    import bcfwriter as W
    hdr = bcf_cases.std_header(extra=['##FORMAT=<ID=FLG,Number=0,Type=Flag,Description="d">'])
    _check(W.bcf_bytes(hdr, bcf_cases.basic_records() * 300))
    _check(W.bcf_bytes(hdr, bcf_cases.basic_records() * 300), tidy=True)


def _parse_csi(d):
    import struct
    assert d[:4] == b"CSI\x01"
    min_shift, depth, l_aux = struct.unpack_from("<iii", d, 4)
    p = 16 + l_aux
    (n_ref,) = struct.unpack_from("<i", d, p); p += 4
    refs = []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", d, p); p += 4
        bins = {}
        for _b in range(n_bin):
            bin_, loff, n_chunk = struct.unpack_from("<IQi", d, p); p += 16
            bins[bin_] = (loff, [struct.unpack_from("<QQ", d, p + 16 * k) for k in range(n_chunk)]); p += 16 * n_chunk
        refs.append(bins)
    (n_no_coor,) = struct.unpack_from("<Q", d, p) if p + 8 <= len(d) else (0,)
    return min_shift, depth, l_aux, refs, n_no_coor


def test_csi_writer_equals_the_reference_index(tmp_path):
    """dhts_bcf_build_index on the reference's vcf_file.bcf against its golden vcf_file.bcf.csi (written by htslib's bcf_index_build): same
    parameters, bins, loffsets, chunks and counts (the file order of bins is khash's and is not part of the format); the wrapped file is
    valid BGZF and serves region queries"""
    import ctypes as C
    import gzip
    import duckhts_amd
    data = _gold("vcf_file.bcf")
    gold = _parse_csi(gzip.decompress(_gold("vcf_file.bcf.csi")))
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); ctx.bgzf_index()
        duckhts_amd.BcfScan(ctx)
        L = duckhts_amd.lib()
        L.dhts_bcf_build_index.restype = C.c_int64
        L.dhts_bcf_build_index.argtypes = [C.c_void_p, C.c_int]
        n = L.dhts_bcf_build_index(ctx.h, 14)
        assert n > 0, L.dhts_error(ctx.h)
        raw = np.zeros(n, np.uint8)
        assert L.dhts_bam_index_bytes(ctx.h, raw.ctypes.data, n) == 0
    finally:
        ctx.close()
    mine = _parse_csi(raw.tobytes())
    assert mine[:3] == gold[:3] == (14, 5, 0) and mine[4] == gold[4]
    assert mine[3] == gold[3]
    L.dhts_bgzf_wrap.restype = C.c_int64
    L.dhts_bgzf_wrap.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    need = L.dhts_bgzf_wrap(raw.ctypes.data, n, None, 0)
    out = np.zeros(need, np.uint8)
    assert L.dhts_bgzf_wrap(raw.ctypes.data, n, out.ctypes.data, need) == need
    assert gzip.decompress(out.tobytes()) == raw.tobytes() and out[-28:].tobytes() == _gold("vcf_file.bcf.csi")[-28:]
    # the written index narrows a region query exactly like the golden one
    for region in ("1:3000150-3000151", "1:3062915-3062915", "4:1-10000000"):
        a = duckhts_amd.read_bcf(data, region=region, index=out.tobytes())
        b = duckhts_amd.read_bcf(data, region=region, index=_gold("vcf_file.bcf.csi"))
        assert orc.bcf_cols_diff(a, b) is None and a["n_rows"] == b["n_rows"]
    # a larger synthetic file: written index vs no index, random regions
    from duckhts_amd import synth
    big = synth.bcf_file(20000, seed=5)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(big); ctx.bgzf_index(); duckhts_amd.BcfScan(ctx)
        n = L.dhts_bcf_build_index(ctx.h, 0)
        assert n > 0, L.dhts_error(ctx.h)
        raw = np.zeros(n, np.uint8); L.dhts_bam_index_bytes(ctx.h, raw.ctypes.data, n)
    finally:
        ctx.close()
    need = L.dhts_bgzf_wrap(raw.ctypes.data, n, None, 0); out = np.zeros(need, np.uint8); L.dhts_bgzf_wrap(raw.ctypes.data, n, out.ctypes.data, need)
    for region in ("chr1:1-2000000", "chr2:100000-100500", "chrX", "chr5:1-1"):
        a = duckhts_amd.read_bcf(big, region=region, index=out.tobytes())
        b = duckhts_amd.read_bcf(big, region=region)
        assert orc.bcf_cols_diff(a, b) is None
