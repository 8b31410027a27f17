"""Rows A4 / A5 on the reference's own SAM-text fixtures (re-encoded as BAM by tests/samtext.py): the read-group / sample columns of
test/data/rg.sam.gz (duckhts.test:164-177) and the standard-tag / auxiliary-tag columns of test/data/aux_tags.sam.gz (duckhts.test:179-185)."""
import pytest

import orc
import samtext


def _rg_expectations(t):
    rg, sm = t["READ_GROUP_ID"], t["SAMPLE_ID"]
    assert t["n_rows"] == 6
    assert sum(x is not None for x in rg) == 4                                   # duckhts.test:164-167
    assert sum(x == b"x1" for x in sm) == 2 and sum(x == b"x2" for x in sm) == 2  # duckhts.test:169-177
    assert rg == [b"x1", b"x2", None, b"x1", b"x2", None] and sm == [b"x1", b"x2", None, b"x1", b"x2", None]
    assert t["QNAME"] == [b"a1", b"b1", b"c1", b"a2", b"b2", b"c2"] and list(t["POS"]) == [1, 1, 1, 11, 11, 11] and t["QUAL"][0] == b"**********"
    assert t["RNEXT"][0] == b"*" and t["CIGAR"][0] == b"10M" and t["SEQ"][3] == b"TTTTTTTTTT"


def test_rg_sam_fixture_oracle():
    _rg_expectations(orc.bam_read(samtext.sam_fixture_as_bam("rg.sam.gz")))


def test_aux_tags_sam_fixture_oracle():
    data = samtext.sam_fixture_as_bam("aux_tags.sam.gz")
    t = orc.bam_read(data)
    assert t["n_rows"] == 1 and t["READ_GROUP_ID"] == [b"x1"] and t["SAMPLE_ID"] == [b"x1"] and t["QUAL"] == [b"!!!!"]
    std = orc.bam_read_std_tags(data)
    by = {c["name"]: orc.bcf_col_py(c) for c in std["cols"]}
    assert by["RG"] == [b"x1"] and by["NM"] == [2]                               # duckhts.test:179-185
    aux = orc.bam_read_aux_map(data, exclude_standard=True)
    keys, vals = (orc.bcf_col_py(c) for c in aux["cols"])
    assert keys == [[b"XZ"]] and vals == [[b"foo"]]                              # map_extract(AUXILIARY_TAGS, 'XZ') = [foo]


@pytest.mark.gpu
def test_sam_fixtures_gpu():
    import duckhts_amd
    data = samtext.sam_fixture_as_bam("rg.sam.gz")
    exp, got = orc.bam_read(data), duckhts_amd.read_bam(data)
    _rg_expectations(got)
    for k in ("QNAME", "CIGAR", "SEQ", "QUAL", "READ_GROUP_ID", "SAMPLE_ID", "RNAME", "RNEXT"):
        assert got[k] == exp[k], k
    data = samtext.sam_fixture_as_bam("aux_tags.sam.gz")
    got = duckhts_amd.read_bam(data, std_tags_cols=list(range(56)), aux_map="exclude_standard")
    std = orc.bam_read_std_tags(data)
    assert orc.bcf_cols_diff({"n_rows": 1, "cols": std["cols"]}, got["tags"]) is None
    by = {c["name"]: orc.bcf_col_py(c) for c in got["tags"]["cols"]}
    assert by["RG"] == [b"x1"] and by["NM"] == [2]
    keys, vals = (orc.bcf_col_py(c) for c in got["aux"]["cols"])
    assert keys == [[b"XZ"]] and vals == [[b"foo"]]
