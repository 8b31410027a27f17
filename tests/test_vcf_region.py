"""read_bcf(region := ...) on bgzipped VCF TEXT and the tabix index writer (the callers and formats either side of
the hot path).

What the reference does (src/bcf_reader.c:904-959, 1296-1345): tbx_index_load3, tbx_itr_querys per region token, tbx_itr_next + vcf_parse1.
The rows are decided by the interval tbx_parse1 computes for each line (htslib tbx.c:96-312, VCF preset: REF length, SVLEN of
<DEL>/<DUP>/<CNV>/<INV>, FORMAT/LEN of gVCF blocks, INFO/END); the oracle restates that rule (oracle/bcf_oracle.c tbx_vcf_end) and stores it as
the text record's rlen, the device encoder does the same, and oracle/region_oracle.py vcf_text_region_rows lists the rows.

Golden indexes (written by htslib itself): third_party/htslib/test/index.vcf.gz.tbi / .csi, test/data/no_contig.vcf.gz.tbi and
test/data/formatcols.vcf.gz.csi.  index.vcf.gz is not shipped; its two indexes pin it down: header in one stored BGZF block (5,096 bytes ->
file offset 5,127 for the records), the 621 records in a second stored block, EOF block at 68,950 -- which is what a level-0 BGZF writer that
flushes after the header produces, and what tests/bamwriter.py writes below."""
import gzip
import os
import random
import struct
import sys
import zlib

import numpy as np
import pytest

import orc
import bamwriter as W
import vcf_text_cases as V

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import region_oracle  # noqa: E402


def _gold(name):
    return open(os.path.join(GOLD, name), "rb").read()


def index_vcf_gz():
    txt = _gold("index.vcf")
    hdr = 0
    for line in txt.splitlines(keepends=True):
        if not line.startswith(b"#"):
            break
        hdr += len(line)
    def stored(raw):                                       # one BGZF block holding one stored DEFLATE block (what level 0 writes)
        body = b"\x01" + struct.pack("<HH", len(raw), len(raw) ^ 0xffff) + raw
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(body) + 8 - 1) + body
                + struct.pack("<II", zlib.crc32(raw), len(raw)))
    data = stored(txt[:hdr]) + stored(txt[hdr:]) + W.EOF_BLOCK
    assert hdr == 5096 and len(data) == 68950 + 28 and gzip.decompress(data) == txt
    return data


def parse_tabix(d):
    """-> dict(kind, min_shift, depth, conf (6 ints), names, refs [{bin: (loff, [(u, v)])}], lin [[...]] (TBI), n_no_coor)"""
    if d[:2] == b"\x1f\x8b":
        d = gzip.decompress(d)
    out = {}
    if d[:4] == b"TBI\x01":
        (n_ref,) = struct.unpack_from("<i", d, 4)
        m, p, out["kind"], out["min_shift"], out["depth"] = d[8:], None, "tbi", 14, 5
        (l_nm,) = struct.unpack_from("<i", m, 24)
        p = 8 + 28 + l_nm
    else:
        assert d[:4] == b"CSI\x01"
        out["min_shift"], out["depth"], l_aux = struct.unpack_from("<iii", d, 4)
        m, out["kind"] = d[16:16 + l_aux], "csi"
        (l_nm,) = struct.unpack_from("<i", m, 24)
        assert l_aux == 28 + l_nm
        (n_ref,) = struct.unpack_from("<i", d, 16 + l_aux)
        p = 16 + l_aux + 4
    out["conf"] = struct.unpack_from("<6i", m, 0)
    out["names"] = m[28:28 + l_nm].split(b"\x00")[:-1]
    refs, lins = [], []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", d, p); p += 4
        bins = {}
        for _b in range(n_bin):
            if out["kind"] == "csi":
                b, loff, nc = struct.unpack_from("<IQi", d, p); p += 16
            else:
                b, nc = struct.unpack_from("<Ii", d, p); p += 8
                loff = 0
            bins[b] = (loff, [struct.unpack_from("<QQ", d, p + 16 * k) for k in range(nc)]); p += 16 * nc
        refs.append(bins)
        if out["kind"] == "tbi":
            (ni,) = struct.unpack_from("<i", d, p); p += 4
            lins.append(list(struct.unpack_from("<%dQ" % ni, d, p))); p += 8 * ni
    out["refs"], out["lin"] = refs, lins
    out["n_no_coor"] = struct.unpack_from("<Q", d, p)[0] if p + 8 <= len(d) else None
    return out


def build_index(data, min_shift):
    """dhts_bcf_build_index on text + dhts_bgzf_wrap -> (raw index bytes, the .tbi / .csi file)"""
    import ctypes as C
    import duckhts_amd
    L = duckhts_amd.lib()
    L.dhts_bcf_build_index.restype = C.c_int64
    L.dhts_bcf_build_index.argtypes = [C.c_void_p, C.c_int]
    L.dhts_bgzf_wrap.restype = C.c_int64
    L.dhts_bgzf_wrap.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); ctx.bgzf_index()
        duckhts_amd.BcfScan(ctx)
        n = L.dhts_bcf_build_index(ctx.h, min_shift)
        assert n > 0, L.dhts_error(ctx.h)
        raw = np.zeros(n, np.uint8)
        assert L.dhts_bam_index_bytes(ctx.h, raw.ctypes.data, n) == 0
    finally:
        ctx.close()
    need = L.dhts_bgzf_wrap(raw.ctypes.data, n, None, 0)
    out = np.zeros(need, np.uint8)
    assert L.dhts_bgzf_wrap(raw.ctypes.data, n, out.ctypes.data, need) == need
    return raw.tobytes(), out.tobytes()


def region_check(data, region, index, tidy=False, **kw):
    import duckhts_amd
    exp = orc.bcf_read(data, tidy)
    reps = exp["n_samples"] if tidy and exp["n_samples"] else 1
    chrom = orc.bcf_col_py(exp["by_name"]["CHROM"])[::reps]
    rows = region_oracle.vcf_text_region_rows(exp, chrom, region_oracle.tabix_names(index), region, reps)
    want = orc.bcf_take_rows(exp, rows)
    got = duckhts_amd.read_bcf(data, tidy=tidy, region=region, index=index, **kw)
    d = orc.bcf_cols_diff(want, got)
    assert d is None, (region, d)
    assert got["status"] == 1 or got["n_rows"] == 0
    return got["n_rows"]


# ---- CPU: the oracle's interval rule and region rows on the reference's files ------------------------------------------------------------
def test_oracle_interval_of_a_line():
    """tbx_parse1, VCF preset: one line per branch of the END rule"""
    hdr = V.HDR[:-1] + ['##INFO=<ID=END,Number=1,Type=Integer,Description="d">', '##INFO=<ID=SVLEN,Number=.,Type=Integer,Description="d">',
                        '##INFO=<ID=XEND,Number=1,Type=Integer,Description="d">', V.HDR[-1]]
    lines = [V.L(pos=100, ref="ACGT"), V.L(pos=200, alt="<DEL>", info="SVLEN=-500"), V.L(pos=300, alt="<DEL>", info="DP=1;END=900;SVLEN=50"),
             V.L(pos=400, alt="T,<DUP:TANDEM>", info="SVLEN=7,300"), V.L(pos=500, info="XEND=900"), V.L(pos=0, ref="AC"),
             V.L(pos=600, alt="<INS>", info="SVLEN=400"), V.L(pos=700, alt="<DELX>", info="SVLEN=400"), V.L(pos=800, info="END=."), V.L(pos=900, info="END=100"),
             V.L(pos=1000, alt="<CNV>", info="SVLEN=."), V.L(pos=1100, info="END=0x500"), V.L(pos=1200, alt="<INV>", info="DP=3;SVLEN=-20;END=1205")]
    t = orc.bcf_read(V.text(lines, hdr=hdr))
    assert t["n_rows"] == len(lines)
    assert list(t["rec"]["pos0"]) == [99, 199, 299, 399, 499, -1, 599, 699, 799, 899, 999, 1099, 1199]
    assert list(t["rec"]["rlen"]) == [4, 500, 601, 300, 1, 3, 1, 1, 1, 1, 1, 0x500 - 1099, 20]


def test_oracle_gvcf_len_blocks():
    hdr = V.SHDR[:-1] + ['##FORMAT=<ID=LEN,Number=1,Type=Integer,Description="d">', '##FORMAT=<ID=MIN_DP,Number=1,Type=Integer,Description="d">', V.SHDR[-1]]
    lines = [V.S("GT:LEN", "0/0:50", "0/0:120", "0/0:7", pos=100).replace("\tT\t", "\t<*>\t"), V.S("GT:MIN_DP:LEN", "0/0:3:9", "0/0", "0/0:4:30", pos=300).replace("\tT\t", "\tT,<NON_REF>\t"),
             V.S("GT:LEN", "0/0:50", "0/0:120", "0/0:7", pos=500)]
    t = orc.bcf_read(V.text(lines, hdr=hdr))
    assert t["n_rows"] == 3 and list(t["rec"]["rlen"]) == [120, 30, 1]


def test_oracle_regions_on_index_vcf():
    """htslib's own indexed VCF: 621 records on 1, 2 and 10; counts by the tabix rule (every record here is one base long)"""
    data = index_vcf_gz()
    tbi = _gold("index.vcf.gz.tbi")
    assert region_oracle.tabix_names(tbi) == region_oracle.tabix_names(_gold("index.vcf.gz.csi")) == ["1", "2", "10"]
    t = orc.bcf_read(data)
    chrom = orc.bcf_col_py(t["by_name"]["CHROM"])
    assert t["n_rows"] == 621
    n = {r: len(region_oracle.vcf_text_region_rows(t, chrom, ["1", "2", "10"], r)) for r in ("1", "2", "10", "1:9999919-9999920", "10:1-10000000", "3", "1,10", ".")}
    assert n["1"] + n["2"] + n["10"] == 621 == n["."] and n["1:9999919-9999920"] == 2 and n["3"] == 0 and n["1,10"] == n["1"] + n["10"]
    g = parse_tabix(tbi)
    assert g["conf"] == (2, 1, 2, 0, ord("#"), 0) and g["names"] == [b"1", b"2", b"10"]
    assert [g["refs"][k][37450][1][1] for k in range(3)] == [(n["1"], 0), (n["2"], 0), (n["10"], 0)]      # the pseudo-bin's record counts


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------
def sv_text(n=30000, seed=7, payload=3000):
    """sorted, three sequences (one without a ##contig line), symbolic alleles with SVLEN / END, gVCF blocks with FORMAT/LEN"""
    rng = random.Random(seed)
    hdr = V.SHDR[:-1] + ['##INFO=<ID=END,Number=1,Type=Integer,Description="d">', '##INFO=<ID=SVLEN,Number=.,Type=Integer,Description="d">',
                        '##FORMAT=<ID=LEN,Number=1,Type=Integer,Description="d">', "##contig=<ID=chr3,length=5000000>", V.SHDR[-1]]
    lines = []
    for chrom in ("chr1", "chr2", "chrUn"):
        pos = 0
        for _ in range(n // 3):
            pos += rng.randint(1, 60)
            k = rng.random()
            if k < 0.05:
                lines.append("\t".join([chrom, str(pos), ".", "A", "<DEL>", "30", "PASS", "SVLEN=-%d" % rng.randint(1, 5000), "GT", "0/1", "0/0", "1/1"]))
            elif k < 0.08:
                lines.append("\t".join([chrom, str(pos), ".", "A", "<DUP>", "30", "PASS", "DP=5;END=%d" % (pos + rng.randint(0, 3000)), "GT", "0/1", "0/0", "1/1"]))
            elif k < 0.12:
                lines.append("\t".join([chrom, str(pos), ".", "A", "<*>", ".", ".", ".", "GT:LEN", "0/0:%d" % rng.randint(1, 800), "0/0:%d" % rng.randint(1, 800), "0/0"]))
            elif k < 0.16:
                lines.append("\t".join([chrom, str(pos), ".", "ACGTACGTAC"[:rng.randint(1, 10)], "A", "50", "PASS", "DP=%d" % rng.randint(1, 99), "GT:GQ", "0/1:5", "0/0:7", "1/1:9"]))
            else:
                lines.append("\t".join([chrom, str(pos), ".", "A", "T,C"[:rng.choice((1, 3))], "50", "PASS", "DP=%d;AF=0.5" % rng.randint(1, 99), "GT:GQ", "0/1:5", "0/0:7", "1/1:9"]))
    return W.bgzf_file(V.text(lines, hdr=hdr), payload=payload)


@pytest.mark.gpu
def test_gpu_records_carry_the_tabix_interval():
    """pos / rlen of every text record = the oracle's (what the predicate and the index writer read)"""
    import duckhts_amd
    data = sv_text(3000)
    exp = orc.bcf_read(data)
    _, tbi = build_index(data, 0)
    t = parse_tabix(tbi)
    assert t["names"] == [b"chr1", b"chr2", b"chrUn"]
    for k, nm in enumerate((b"chr1", b"chr2", b"chrUn")):
        chrom = orc.bcf_col_py(exp["by_name"]["CHROM"])
        assert t["refs"][k][37450][1][1] == (sum(1 for c in chrom if c == nm), 0)
    got = duckhts_amd.read_bcf(data)
    assert orc.bcf_cols_diff(exp, got) is None


@pytest.mark.gpu
def test_gpu_tabix_writer_equals_htslib_indexes():
    """TBI and CSI of index.vcf.gz, TBI of no_contig.vcf.gz, CSI of formatcols.vcf.gz: header, names, every bin with its chunks (and loffset),
    the linear index and the counts equal the files htslib wrote"""
    for data, gold, ms in ((index_vcf_gz(), "index.vcf.gz.tbi", 0), (index_vcf_gz(), "index.vcf.gz.csi", 14), (_gold("no_contig.vcf.gz"), "no_contig.vcf.gz.tbi", 0),
                           (_gold("formatcols.vcf.gz"), "formatcols.vcf.gz.csi", 14)):
        raw, wrapped = build_index(data, ms)
        mine, ref = parse_tabix(raw), parse_tabix(_gold(gold))
        assert gzip.decompress(wrapped) == raw
        for k in ("kind", "min_shift", "depth", "conf", "names", "n_no_coor", "lin"):
            assert mine[k] == ref[k], (gold, k, mine[k], ref[k])
        assert mine["refs"] == ref["refs"], gold


@pytest.mark.gpu
def test_gpu_regions_on_index_vcf():
    data = index_vcf_gz()
    for index in (_gold("index.vcf.gz.tbi"), _gold("index.vcf.gz.csi"), build_index(data, 0)[1], build_index(data, 14)[1]):
        assert region_check(data, "1:9999919-9999920", index) == 2
        assert region_check(data, "2", index) + region_check(data, "1", index) + region_check(data, "10", index) == 621
        assert region_check(data, "10:1-10000000,3,1:10000000-10000100,.", index, max_blocks=1) > 621
        assert region_check(data, "3", index) == 0 and region_check(data, "chr1", index) == 0
    assert region_check(_gold("no_contig.vcf.gz"), "chr1:1-1000", _gold("no_contig.vcf.gz.tbi")) == 1        # a sequence only the index (and the line) names
    assert region_check(_gold("no_contig.vcf.gz"), "chr1:2-3", _gold("no_contig.vcf.gz.tbi")) == 0
    assert region_check(_gold("formatcols.vcf.gz"), "1", _gold("formatcols.vcf.gz.csi"), tidy=True) == 3


@pytest.mark.gpu
def test_gpu_regions_on_a_many_block_file():
    """windows that start and end inside the file, intervals longer than REF (SVLEN / END / LEN), a sequence without a ##contig line"""
    import duckhts_amd
    data = sv_text()
    _, tbi = build_index(data, 0)
    _, csi = build_index(data, 12)
    rng = random.Random(3)
    regions = ["chr1:1-1000", "chr2:100000-101000", "chrUn", "chrUn:250000-260000,chr1:5-6", "chr3", "chr2:299000-", "nosuch,chr1:150000-150100"]
    for _ in range(6):
        b = rng.randint(1, 300000)
        regions.append("%s:%d-%d" % (rng.choice(("chr1", "chr2", "chrUn")), b, b + rng.randint(0, 4000)))
    total = 0
    for k, rg in enumerate(regions):
        total += region_check(data, rg, tbi if k % 2 == 0 else csi, max_blocks=(0, 3, 1)[k % 3])
    assert total > 1000
    assert region_check(data, "chr2:1000-200000", tbi, tidy=True, max_blocks=7) > 0
    # the window is narrower than the file (records with long intervals sit in high bins whose chunks are spread over the sequence, so for a
    # file this small the covering window is the sequence itself)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(data); n_blocks = ctx.bgzf_index()
        sc = duckhts_amd.BcfScan(ctx)
        assert sc.set_region("chr2:100000-101000") and sc.load_index(tbi)
        nw, nb = ctx.scan_window_stats()
        assert 0 < nb <= n_blocks // 3 + 2, (nb, n_blocks)
        assert sc.set_region("nosuch") and not sc.load_index(tbi)
    finally:
        ctx.close()
    with pytest.raises(duckhts_amd.DhtsError, match="needs the tabix index"):
        duckhts_amd.read_bcf(data, region="chr1:1-10")


@pytest.mark.gpu
def test_gpu_region_on_vcf_text_through_the_table_function(tmp_path):
    """the table function finds <file>.tbi / .csi itself (bcf_reader.c:904-916) and chains the regions"""
    from test_duckdb_surface import run_host
    data = sv_text(9000)
    fn = os.path.join(str(tmp_path), "sv.vcf.gz")
    open(fn, "wb").write(data)
    exp = orc.bcf_read(data)
    chrom = orc.bcf_col_py(exp["by_name"]["CHROM"])
    for ext, ms in ((".tbi", 0), (".csi", 14)):
        open(fn + ext, "wb").write(build_index(data, ms)[1])
        for rg in ("chr2:1000-30000", "chrUn:1-2000,chr1:100-5000,zzz", "chr3"):
            want = len(region_oracle.vcf_text_region_rows(exp, chrom, ["chr1", "chr2", "chrUn"], rg))
            rc, out, _ = run_host(fn, named=[("region", rg)], fn="read_bcf")
            assert rc == 0 and ("rows=%d " % want) in out, (rg, want, out)
        os.remove(fn + ext)


@pytest.mark.gpu
@pytest.mark.parametrize("gap_mb", ["0", "0.01"])
def test_gpu_disjoint_index_windows(gap_mb, monkeypatch):
    """DHTS_WINDOW_GAP_MB=0 makes every merged index chunk a window of its own (the default, 32 MB, merges everything in files this small):
    each window is cut exactly at its end, so no row comes twice, and the rows are the oracle's -- text and binary"""
    import duckhts_amd
    from duckhts_amd import synth
    monkeypatch.setenv("DHTS_WINDOW_GAP_MB", gap_mb)
    data = sv_text(450000, seed=9, payload=65280)                  # (large enough that bins keep their own chunks: compress_binning folds a bin of < 64 KiB into its parent)
    for ms in (0, 10):
        _, idx = build_index(data, ms)
        ctx = duckhts_amd.Context(0)
        nws = []
        try:
            ctx.open(data); ctx.bgzf_index()
            sc = duckhts_amd.BcfScan(ctx)
            for rg in ("chr1:1000000-1000100", "chr2:2000000-2000500", "chrUn:50000-51000", "chr1:3900000-3900010"):
                assert sc.set_region(rg) and sc.load_index(idx)
                nws.append(ctx.scan_window_stats()[0])
        finally:
            ctx.close()
        print("index windows", gap_mb, ms, nws)
        if gap_mb == "0":
            assert max(nws) > 1, nws                                                    # long intervals sit in high bins: several chunks
        for k, rg in enumerate(("chr1:1000000-1250000", "chr2:1-50,chr2:4200000-", "chrUn:1000000-1000100", "chr2:1500000-1500001,chr1:1500000-1500001")):
            region_check(data, rg, idx, max_blocks=(0, 1, 5)[(k + ms) % 3])
    import test_gpu_bcf as TB
    big = synth.bcf_file(40000, seed=11, payload=2000)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(big); ctx.bgzf_index(); duckhts_amd.BcfScan(ctx)
        import ctypes as C
        L = duckhts_amd.lib()
        L.dhts_bcf_build_index.restype = C.c_int64; L.dhts_bcf_build_index.argtypes = [C.c_void_p, C.c_int]
        n = L.dhts_bcf_build_index(ctx.h, 12)
        raw = np.zeros(n, np.uint8); L.dhts_bam_index_bytes(ctx.h, raw.ctypes.data, n)
        csi = ctx.bgzf_compress(raw.tobytes())
    finally:
        ctx.close()
    for rg in ("chr1:1-3000000", "chr2:100000-5000000,chr1:1-10", "chrX", "chr5:1000000-1000001"):
        for mb in (0, 2):
            TB._region_check(big, rg, index=csi, max_blocks=mb)


@pytest.mark.gpu
def test_gpu_region_queries_stage_only_their_windows(tmp_path, monkeypatch):
    """read_bcf(region := ...) through the table function stages the header blocks and the regions' index windows, not the file (the reference
    seeks to the chunks): same rows as with DHTS_SPARSE=0, a fraction of the bytes resident; text and binary"""
    import duckhts_amd
    from duckhts_amd import synth
    from test_duckdb_surface import run_host
    L = duckhts_amd.lib()
    monkeypatch.setenv("DHTS_WINDOW_GAP_MB", "0.05")             # (the default, 32 MB, would merge every window of files this small)
    data = sv_text(450000, seed=9, payload=65280)
    _, tbi = build_index(data, 0)
    big = synth.bcf_file(60000, seed=11)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(big); ctx.bgzf_index(); duckhts_amd.BcfScan(ctx)
        import ctypes as C
        L.dhts_bcf_build_index.restype = C.c_int64; L.dhts_bcf_build_index.argtypes = [C.c_void_p, C.c_int]
        n = L.dhts_bcf_build_index(ctx.h, 14)
        raw = np.zeros(n, np.uint8); L.dhts_bam_index_bytes(ctx.h, raw.ctypes.data, n)
        csi = ctx.bgzf_compress(raw.tobytes())
    finally:
        ctx.close()
    for name, blob, index, ext, regions in (("t.vcf.gz", data, tbi, ".tbi", "chr1:1000000-1000100,chrUn:50000-51000,nosuch,chr2:4000000-"), ("b.bcf", big, csi, ".csi", "chr2:100000-5000000,chr1:1-10,chrX:1-3000000")):
        fn = os.path.join(str(tmp_path), name)
        open(fn, "wb").write(blob); open(fn + ext, "wb").write(index)
        # C ABI: segments from a header context, a second context that holds only them
        hdr = duckhts_amd.Context(0)
        try:
            hdr.open(fn); hdr.bgzf_index(); duckhts_amd.BcfScan(hdr)
            seg = hdr.bcf_region_segments(regions, index)
            assert seg is not None and len(seg[1]) >= 2
            assert hdr.bcf_region_segments("chr1,.", index) is None                    # "." needs the whole file
        finally:
            hdr.close()
        c2 = duckhts_amd.Context(0)
        try:
            c2.open_segments(fn, *seg); c2.bgzf_index()
            assert L.dhts_resident_bytes(c2.h) < len(blob) // 3, (L.dhts_resident_bytes(c2.h), len(blob))
            sc = duckhts_amd.BcfScan(c2)
            total = 0
            for rg in regions.split(","):
                if not sc.set_region(rg) or not sc.load_index(index):
                    continue
                while True:
                    b = sc.next_batch(3)
                    total += b.n_rows
                    if b.status != 0:
                        assert b.status == 1, b.status
                        break
        finally:
            c2.close()
        want = duckhts_amd.read_bcf(blob, region=regions, index=index)["n_rows"]
        assert total == want and want > 0
        # the table function: same rows with and without sparse staging
        outs = []
        for env in ({}, {"DHTS_SPARSE": "0"}):
            rc, out, _ = run_host(fn, named=[("region", regions)], fn="read_bcf", env=env)
            assert rc == 0, out
            outs.append(out.split("rows=")[1].split()[0])
        assert outs[0] == outs[1] == str(want), (outs, want)
