"""bgzip / bgunzip / bam_index / bcf_index / tabix_index through the DuckDB table-function surface (mini host), the way the reference's
duckhts.test calls them (:261-274) -- same names, named parameters, one result row (src/bgzip.c:74-86, src/hts_index_builder.c:70-79) and
error strings; the files they write are checked against htslib's own golden indexes and by reading them back."""
import gzip
import os
import shutil
import struct

import pytest

import orc
import test_vcf_region as TR
from test_duckdb_surface import parse_chunks, run_host

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def one_row(fn, path, named=()):
    rc, out, chunks = run_host(path, named=named, fn=fn)
    assert rc == 0 and "rows=1 " in out, out
    schema, ch = parse_chunks(chunks)
    (n, cols), = ch
    vals = []
    for (_t, _valid, v) in cols:
        x = v[0]
        vals.append(x.decode() if isinstance(x, (bytes, bytearray)) else int(x))
    return [s[0] for s in schema], vals


def parse_bai(d):
    assert d[:4] == b"BAI\x01"
    (n_ref,) = struct.unpack_from("<i", d, 4); p = 8
    refs = []
    for _ in range(n_ref):
        (nb,) = struct.unpack_from("<i", d, p); p += 4
        bins = {}
        for _b in range(nb):
            b, nc = struct.unpack_from("<Ii", d, p); p += 8
            bins[b] = [struct.unpack_from("<QQ", d, p + 16 * k) for k in range(nc)]; p += 16 * nc
        (ni,) = struct.unpack_from("<i", d, p); p += 4
        refs.append((bins, list(struct.unpack_from("<%dQ" % ni, d, p)))); p += 8 * ni
    return refs, d[p:]


@pytest.mark.gpu
def test_gpu_bgzip_bgunzip_table_functions(tmp_path):
    import vep_cases
    raw = vep_cases.fixture_text().encode()
    src = os.path.join(str(tmp_path), "t.vcf")
    open(src, "wb").write(raw)
    names, vals = one_row("bgzip", src, named=[("keep", "true")])                              # default output: <input>.gz (bgzip.c:36-42)
    assert names == ["success", "output_path", "bytes_in", "bytes_out"]
    z = open(src + ".gz", "rb").read()
    assert vals == [1, src + ".gz", len(raw), len(z)] and gzip.decompress(z) == raw and os.path.exists(src)
    rc, out, _ = run_host(src, fn="bgzip")
    assert rc == 3 and out == f"ERROR bind: bgzip: output '{src}.gz' already exists (use overwrite := TRUE to replace)"
    names, vals = one_row("bgzip", src, named=[("output_path", src + ".l0.gz"), ("level", "0"), ("keep", "false"), ("overwrite", "true")])
    assert vals[2] == len(raw) and vals[3] == len(raw) + 31 * ((len(raw) + 65279) // 65280) + 28 and not os.path.exists(src)     # level 0 stores; keep := FALSE unlinks
    names, vals = one_row("bgunzip", src + ".gz")                                             # default output: without .gz (bgzip.c:44-58)
    assert vals == [1, src, len(z), len(raw)] and open(src, "rb").read() == raw
    names, vals = one_row("bgunzip", src + ".l0.gz", named=[("output_path", src + ".back")])
    assert open(src + ".back", "rb").read() == raw
    rc, out, _ = run_host(os.path.join(str(tmp_path), "nope"), fn="bgzip")
    assert rc == 3 and out.startswith("ERROR bind: bgzip: cannot open input ")
    # the reference's own use: bgzip, index, region query (duckhts.test:261-283, there on a BED file)
    one_row("tabix_index", src + ".gz", named=[("preset", "vcf"), ("threads", "1")])
    rc, out, _ = run_host(src + ".gz", named=[("region", "1:10000-20000")], fn="read_bcf")
    exp = orc.bcf_read(raw)
    import region_oracle
    chrom = orc.bcf_col_py(exp["by_name"]["CHROM"])
    want = len(region_oracle.vcf_text_region_rows(exp, chrom, region_oracle.tabix_names(open(src + ".gz.tbi", "rb").read()), "1:10000-20000"))
    assert rc == 0 and ("rows=%d " % want) in out and want > 0, out


@pytest.mark.gpu
def test_gpu_index_table_functions(tmp_path):
    d = str(tmp_path)
    # bcf_index on a .bcf: CSI, min_shift 14 by default (hts_index_builder.c:182), equal to the golden index of the same file
    bcf = os.path.join(d, "vcf_file.bcf"); shutil.copy(os.path.join(GOLD, "vcf_file.bcf"), bcf)
    names, vals = one_row("bcf_index", bcf)
    assert names == ["success", "index_path", "index_format"] and vals == [1, bcf + ".csi", "CSI"]
    import test_gpu_bcf as TB
    assert TB._parse_csi(gzip.decompress(open(bcf + ".csi", "rb").read())) == TB._parse_csi(gzip.decompress(open(os.path.join(GOLD, "vcf_file.bcf.csi"), "rb").read()))
    rc, out, _ = run_host(bcf, named=[("min_shift", "0")], fn="bcf_index")
    assert rc == 3 and out == f"ERROR bind: bcf_index: failed to build index for {bcf} (error -1)"        # "TBI indices for BCF files are not supported" (vcf.c:4712-4714)
    # bcf_index / tabix_index on bgzipped VCF text: TBI by default, CSI with min_shift; equal to htslib's own indexes of index.vcf.gz
    vz = os.path.join(d, "index.vcf.gz"); open(vz, "wb").write(TR.index_vcf_gz())
    assert one_row("bcf_index", vz)[1] == [1, vz + ".tbi", "TBI"]
    mine, ref = TR.parse_tabix(open(vz + ".tbi", "rb").read()), TR.parse_tabix(open(os.path.join(GOLD, "index.vcf.gz.tbi"), "rb").read())
    assert mine == ref
    assert one_row("tabix_index", vz, named=[("min_shift", "14"), ("index_path", os.path.join(d, "x.csi"))])[1] == [1, os.path.join(d, "x.csi"), "CSI"]
    assert TR.parse_tabix(open(os.path.join(d, "x.csi"), "rb").read()) == TR.parse_tabix(open(os.path.join(GOLD, "index.vcf.gz.csi"), "rb").read())
    # the other presets and custom columns: the reference's own tabix fixtures (gff: generic 1/4/5; two TSVs with custom columns, a skipped
    # header line and '#' lines; bgzipped SAM with the sam preset: interval end from the CIGAR)
    for data, gold, named in (("gff_file.gff.gz", "gff_file.gff.gz.tbi", [("preset", "gff")]),
                              ("header_tabix.tsv.gz", "header_tabix.tsv.gz.tbi", [("preset", "gff"), ("seq_col", "1"), ("start_col", "2"), ("end_col", "2"), ("skip_lines", "1")]),
                              ("meta_tabix.tsv.gz", "meta_tabix.tsv.gz.tbi", [("preset", "gff"), ("seq_col", "1"), ("start_col", "2"), ("end_col", "2"), ("skip_lines", "1"), ("comment_char", "#")]),
                              ("rg.sam.gz", "rg.sam.gz.tbi", [("preset", "sam")])):
        f = os.path.join(d, data); shutil.copy(os.path.join(GOLD, data), f)
        assert one_row("tabix_index", f, named=named)[1] == [1, f + ".tbi", "TBI"]
        mine, ref = TR.parse_tabix(open(f + ".tbi", "rb").read()), TR.parse_tabix(open(os.path.join(GOLD, gold), "rb").read())
        assert mine == ref, (data, mine, ref)
    bed = os.path.join(d, "t.bed")
    open(bed, "wb").write(b"#track\nchrA\t0\t10\tx\nchrA\t5\t300000\ty\nchrB\t7\t8\tz\n")
    one_row("bgzip", bed)
    assert one_row("tabix_index", bed + ".gz", named=[("preset", "bed"), ("min_shift", "12")])[1] == [1, bed + ".gz.csi", "CSI"]
    t = TR.parse_tabix(open(bed + ".gz.csi", "rb").read())
    assert t["conf"] == (0x10000, 1, 2, 3, ord("#"), 0) and t["names"] == [b"chrA", b"chrB"] and t["min_shift"] == 12
    assert [t["refs"][k][(1 << (3 * t["depth"] + 3)) // 7 + 1][1][1] for k in range(2)] == [(2, 0), (1, 0)]          # the pseudo-bin: records per sequence
    open(bed, "wb").write(b"chrA\tx\t10\n")
    one_row("bgzip", bed, named=[("overwrite", "true")])
    rc, out, _ = run_host(bed + ".gz", named=[("preset", "bed")], fn="tabix_index")
    assert rc == 3 and out == f"ERROR bind: tabix_index: failed to build index for {bed}.gz (error -1)"              # "expected int" (tbx.c:136)
    # bam_index: BAI equal to the golden one; CSI serves region queries like the BAI
    bam = os.path.join(d, "range.bam"); shutil.copy(os.path.join(GOLD, "range.bam"), bam)
    assert one_row("bam_index", bam)[1] == [1, bam + ".bai", "BAI"]
    assert parse_bai(open(bam + ".bai", "rb").read()) == parse_bai(open(os.path.join(GOLD, "range.bam.bai"), "rb").read())
    assert one_row("bam_index", bam, named=[("min_shift", "14"), ("index_path", bam + ".x.csi")])[1] == [1, bam + ".x.csi", "CSI"]
    csi = TB._parse_csi(gzip.decompress(open(bam + ".x.csi", "rb").read()))
    assert csi[0] == 14 and csi[2] == 0
    counts = []
    for idx in (bam + ".bai", bam + ".x.csi"):
        rc, out, _ = run_host(bam, named=[("region", "CHROMOSOME_I:1-2000,CHROMOSOME_V"), ("index_path", idx)], fn="read_bam")
        assert rc == 0, out
        counts.append(out.split("rows=")[1].split()[0])
    assert counts[0] == counts[1] and int(counts[0]) > 0
    # not a BAM: sam_index_build3's "format not indexable"
    rc, out, _ = run_host(vz, fn="bam_index")
    assert rc == 3 and out == f"ERROR bind: bam_index: failed to build index for {vz} (error -3)"
    rc, out, _ = run_host(os.path.join(d, "missing.bam"), fn="bam_index")
    assert rc == 3 and out == f"ERROR bind: bam_index: failed to build index for {os.path.join(d, 'missing.bam')} (error -2)"
