"""Edge cases of the writers: empty inputs, exact block multiples, files without records, a last line without a newline."""
import gzip
import os

import pytest

import orc
import test_vcf_region as TR
import vcf_text_cases as V
from test_bgzip import EOF_BLOCK, check_file
from test_duckdb_surface import run_host


@pytest.mark.gpu
def test_gpu_empty_and_exact_inputs(tmp_path):
    import duckhts_amd
    d = str(tmp_path)
    ctx = duckhts_amd.Context(0)
    try:
        src, dst, back = os.path.join(d, "e"), os.path.join(d, "e.gz"), os.path.join(d, "e.back")
        open(src, "wb").close()
        assert ctx.bgzip_file(src, dst) == (0, 28) and open(dst, "rb").read() == EOF_BLOCK           # bgzip of nothing: the EOF block alone
        assert ctx.bgunzip_file(dst, back) == (28, 0) and os.path.getsize(back) == 0
        for n in (65280, 2 * 65280, 65280 * 3 + 1):
            raw = (b"0123456789abcdef" * (n // 16 + 1))[:n]
            open(src, "wb").write(raw)
            nin, nout = ctx.bgzip_file(src, dst)
            z = open(dst, "rb").read()
            assert (nin, nout) == (n, len(z))
            check_file(raw, z)
            assert ctx.bgunzip_file(dst, back)[1] == n and open(back, "rb").read() == raw
        with pytest.raises(duckhts_amd.DhtsError, match="cannot open output"):
            ctx.bgzip_file(src, os.path.join(d, "no", "such", "dir", "x.gz"))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_index_of_a_file_without_records(tmp_path):
    """tbx_index on a header-only file: hts_idx_init at the end of the header, no sequences (tbx.c:497)"""
    import duckhts_amd
    ctx = duckhts_amd.Context(0)
    z = ctx.bgzf_compress(V.text([]))
    ctx.close()
    raw, tbi = TR.build_index(z, 0)
    t = TR.parse_tabix(tbi)
    assert t["names"] == [] and t["refs"] == [] and t["conf"] == (2, 1, 2, 0, ord("#"), 0)
    got = duckhts_amd.read_bcf(z, region="chr1:1-100", index=tbi)
    assert got["n_rows"] == 0


@pytest.mark.gpu
def test_gpu_region_on_text_without_a_final_newline_and_at_the_file_ends():
    import duckhts_amd
    lines = [V.L(chrom="chr1", pos=10 * i + 1, info="DP=%d" % (i % 50)) for i in range(20000)] + [V.L(chrom="chr2", pos=5 + i) for i in range(3000)]
    ctx = duckhts_amd.Context(0)
    z = ctx.bgzf_compress(V.text(lines, last_eol=False))
    ctx.close()
    _, tbi = TR.build_index(z, 0)
    for rg, want in (("chr1:1-1", 1), ("chr1:1-10", 1), ("chr1:1-11", 2), ("chr2:3004-3004", 1), ("chr2:3004-", 1), ("chr2:3005", 0), ("chr2", 3000), ("chr1:199991-", 1), ("chr1:199992-", 0)):
        assert TR.region_check(z, rg, tbi, max_blocks=1) == want, rg


@pytest.mark.gpu
def test_gpu_tabix_of_a_tsv_with_unsorted_or_broken_lines(tmp_path):
    d = str(tmp_path)
    def make(name, body):
        f = os.path.join(d, name)
        open(f, "wb").write(body)
        rc, out, _ = run_host(f, fn="bgzip")
        assert rc == 0, out
        return f + ".gz"
    ok = make("ok.bed", b"c1\t5\t9\nc1\t5\t6\nc1\t7\t8\nc2\t0\t1\n")
    rc, out, _ = run_host(ok, named=[("preset", "bed")], fn="tabix_index")
    assert rc == 0, out
    unsorted = make("unsorted.bed", b"c1\t50\t90\nc1\t5\t6\n")
    rc, out, _ = run_host(unsorted, named=[("preset", "bed")], fn="tabix_index")
    assert rc == 3 and out.endswith("(error -1)")                                     # "Unsorted positions" (hts_idx_push)
    split = make("split.bed", b"c1\t5\t9\nc2\t5\t6\nc1\t7\t8\n")
    rc, out, _ = run_host(split, named=[("preset", "bed")], fn="tabix_index")
    assert rc == 3 and out.endswith("(error -1)")                                     # "Chromosome blocks not continuous"
    short = make("short.bed", b"c1\t5\n")
    rc, out, _ = run_host(short, named=[("preset", "bed")], fn="tabix_index")
    assert rc == 0                                                                    # (no end column: end = beg + 1, tbx.c:131-141)
    empty_line = make("empty.bed", b"c1\t5\t9\n\nc1\t7\t8\n")
    rc, out, _ = run_host(empty_line, named=[("preset", "bed")], fn="tabix_index")
    assert rc == 3 and out.endswith("(error -1)")                                     # a line without columns does not parse
