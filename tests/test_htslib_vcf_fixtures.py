"""htslib's formatmissing.vcf (+ formatmissing-out.vcf) and vcf44_1.vcf (+ vcf44_1.expected), test.pl:1184-1201, read as TEXT by read_bcf.

formatmissing: a FORMAT column of "." with "." samples parses (vcf_parse_format, vcf.c:3137-3985) to a record without FORMAT fields: every
FORMAT_<ID>_<sample> column of the header's schema is NULL, and htslib writes the line back unchanged (the -out file equals the input).
vcf44_1: VCFv4.4 genotypes with explicit and implicit phasing of the FIRST allele ("/0|1", "|0", ...).  The reference's GT writer
(bcf_reader.c:1904-1957) prints an allele's separator only in front of alleles 2.., so the table shows every genotype without its leading
"/" or "|" -- whichever way htslib's updatephasing (vcf.c:1985-2029) settles the first allele's phase bit, which is what vcf44_1.expected
records: stripping the leading separator from the input file and from the expected file must give the same strings, and the table's."""
import os

import pytest

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "htslib_vcf")


def _rows(name):
    out = []
    for l in open(os.path.join(GOLD, name)).read().split("\n"):
        if l and not l.startswith("#"):
            out.append(l.split("\t"))
    return out


def _strip(gt):
    return gt[1:] if gt[:1] in "/|" and len(gt) > 1 else gt


def _check_formatmissing(t):
    assert open(os.path.join(GOLD, "formatmissing.vcf")).read() == open(os.path.join(GOLD, "formatmissing-out.vcf")).read()
    by = {c["name"]: orc.bcf_col_py(c) for c in t["cols"]}
    assert t["n_rows"] == 1 and by["CHROM"] == [b"1"] and by["POS"] == [100] and by["ID"] == [b"a"] and by["REF"] == [b"A"] and by["ALT"] == [[b"T"]]
    assert by["QUAL"] == [None] and by["FILTER"] == [[b"PASS"]]          # FILTER "." has no entries: the reference shows ['PASS'] (bcf_reader.c:1443-1447)
    for s in ("S1", "S2", "S3"):
        assert by[f"FORMAT_S_{s}"] == [None]


def _check_vcf44(t):
    src, exp = _rows("vcf44_1.vcf"), _rows("vcf44_1.expected")
    assert len(src) == len(exp) == 28
    want = [[_strip(r[9]).encode(), _strip(r[10]).encode()] for r in src]
    assert want == [[_strip(r[9]).encode(), _strip(r[10]).encode()] for r in exp]          # htslib's own round trip changes leading separators only
    by = {c["name"]: orc.bcf_col_py(c) for c in t["cols"]}
    assert t["n_rows"] == 28 and by["POS"] == [int(r[1]) for r in src] and by["ID"] == [r[2].encode() for r in src]
    assert by["FORMAT_GT_HG00096"] == [w[0] for w in want]
    assert by["FORMAT_GT_HG00097"] == [w[1] for w in want]


def test_formatmissing_oracle():
    _check_formatmissing(orc.bcf_read(open(os.path.join(GOLD, "formatmissing.vcf"), "rb").read()))


def test_vcf44_phasing_oracle():
    _check_vcf44(orc.bcf_read(open(os.path.join(GOLD, "vcf44_1.vcf"), "rb").read()))


@pytest.mark.gpu
def test_htslib_vcf_fixtures_gpu():
    import duckhts_amd
    for name, chk in (("formatmissing.vcf", _check_formatmissing), ("vcf44_1.vcf", _check_vcf44)):
        data = open(os.path.join(GOLD, name), "rb").read()
        got = duckhts_amd.read_bcf(data, device=0)
        chk(got)
        assert orc.bcf_cols_diff(orc.bcf_read(data), got) is None
