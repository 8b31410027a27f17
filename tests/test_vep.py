"""VEP_* columns of read_bcf (src/vep_parser.c, src/bcf_reader.c:582-603, 1463-1541).

CPU: the oracle (oracle/bcf_oracle.c emit_vep) pinned on the reference's own fixture test/data/test_vep.vcf -- the two expectations
of test/sql/duckhts.test:107-121 and an independent Python reading of the VCF text for every record, transcript and field -- plus
hand-derived edge cases following the cited reference lines.  GPU: the HIP path against the oracle, bit for bit.
"""
import math

import numpy as np
import pytest

import orc
import vep_cases

INT_MIN = -(1 << 31)


@pytest.fixture(scope="module")
def fixture():
    data, fields, rows = vep_cases.fixture_bcf()
    return data, fields, rows, orc.bcf_read(data)


def test_fixture_schema(fixture):
    data, fields, rows, t = fixture
    names = [c["name"] for c in t["cols"]]
    # 7 core columns, one VEP_<field> per field of the Description's "Format:" list, then the INFO columns (bcf_reader.c:582-603)
    assert names[:7] == ["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER"]
    assert names[7:7 + len(fields)] == ["VEP_" + f for f in fields] and names[7 + len(fields):] == ["INFO_CSQ", "INFO_AF"]
    types = {c["name"]: (orc.BCF_TYPES[c["type"]], c["is_list"]) for c in t["cols"]}
    # vep_infer_type (vep_parser.c:70-90)
    assert types["VEP_DISTANCE"] == ("INTEGER", 1) and types["VEP_STRAND"] == ("INTEGER", 1) and types["VEP_Consequence"] == ("VARCHAR", 1)
    assert types["VEP_gnomAD_AF"] == ("FLOAT", 1) and types["VEP_MAX_AF"] == ("FLOAT", 1) and types["VEP_Allele"] == ("VARCHAR", 1)
    for f in fields:
        exp = "INTEGER" if f in ("DISTANCE", "STRAND", "TSL", "GENE_PHENO", "HGVS_OFFSET") or f.startswith("MOTIF_POS") else \
              "VARCHAR" if f in ("Consequence", "FLAGS", "CLIN_SIG") else \
              "FLOAT" if ("_AF" in f or "AF_" in f or "MOTIF_SCORE_CHANGE" in f or f.startswith("SpliceAI_pred_DS_")) else "VARCHAR"
        assert types["VEP_" + f] == (exp, 1), f
    assert t["n_rows"] == 802 and t["status"] == 0


def test_fixture_duckhts_sql_expectations(fixture):
    data, fields, rows, t = fixture
    col = {n: orc.bcf_col_py(t["by_name"][n]) for n in ("VEP_Allele", "VEP_SYMBOL")}
    assert col["VEP_Allele"][0] is not None                                    # duckhts.test:108-113
    assert (col["VEP_Allele"][0][0], col["VEP_SYMBOL"][0][0]) == (b"T", b"WASH7P")   # duckhts.test:116-121


def test_fixture_every_cell_against_the_vcf_text(fixture):
    data, fields, rows, t = fixture
    nf = len(fields)
    split = [vep_cases.py_split(csq, nf) for _, csq in rows]
    assert sum(len(s) for s in split if s) == 2219      # transcripts in the fixture
    for k, f in enumerate(fields):
        c = t["by_name"]["VEP_" + f]
        got = orc.bcf_col_py(c)
        kind = orc.BCF_TYPES[c["type"]]
        for r, s in enumerate(split):
            if s is None:
                assert got[r] is None
                continue
            exp = [tr[k] for tr in s]
            if kind == "VARCHAR":
                assert got[r] == [None if x is None else x.encode() for x in exp], (f, r)
            elif kind == "INTEGER":
                assert got[r] == [None if x is None else int(x) for x in exp], (f, r)
            else:
                def flt(x):                               # MAX_AF_POPS is typed FLOAT by its name but holds population names: NaN, not NULL
                    try:
                        return float(np.float32(float(x)))
                    except ValueError:
                        return math.nan
                e = [None if x is None else flt(x) for x in exp]
                assert len(got[r]) == len(e) and all((a is None and b is None) or (a is not None and b is not None and (a == b or (math.isnan(a) and math.isnan(b))))
                                                     for a, b in zip(got[r], e)), (f, r)


def _case(name):
    return {n: (d, t) for n, d, t in vep_cases.edge_cases()}[name]


def test_edge_values():
    d, tidy = _case("csq_values")
    t = orc.bcf_read(d, tidy)
    col = {c["name"][4:]: orc.bcf_col_py(c) for c in t["cols"] if c["name"].startswith("VEP_")}
    assert list(col) == vep_cases.FIELDS
    assert col["Allele"][1] == [b"T", b"G"] and col["DISTANCE"][1] == [3, -4] and col["STRAND"][1] == [1, 1] and col["FLAGS"][1] == [None, b"f"]
    assert col["gnomAD_AF"][1][0] == 0.5 and math.copysign(1, col["gnomAD_AF"][1][1]) == -1 and col["MAX_AF"][1] == [5.0, math.inf] and math.isnan(col["SpliceAI_pred_DS_AG"][1][1])
    assert col["Allele"][2] == [b"T"] and col["Consequence"][2] == [b"only_two"] and col["SYMBOL"][2] == [None]      # strtok_r skips empty pieces; short transcript
    assert all(v[3] is None for v in col.values())                                                                    # ",,,": no transcript -> NULL row (bcf_reader.c:1532-1537)
    assert all(v[4] == [None] for v in col.values())                                                                  # "." is one transcript whose first field is missing
    # trimming and failed conversions: the element stays valid with INT32_MIN / NaN (vep_parser.c:207-235, bcf_reader.c:1497-1518)
    assert col["Allele"][5] == [b"T"] and col["Consequence"][5] == [None] and col["SYMBOL"][5] == [None] and col["NOTE"][5] == [b"spaced  out"]
    assert col["DISTANCE"][5] == [INT_MIN] and col["STRAND"][5] == [INT_MIN] and col["MOTIF_POS"][5] == [-1] and col["FLAGS"][5] == [None]
    assert math.isnan(col["gnomAD_AF"][5][0]) and col["MAX_AF"][5] == [16.0] and col["SpliceAI_pred_DS_AG"][5] == [math.inf]
    assert col["DISTANCE"][6] == [0, None] and col["STRAND"][6] == [INT_MIN, None] and col["gnomAD_AF"][6] == [0.0, None] and col["NOTE"][6] == [None, None]
    assert col["DISTANCE"][7] == [7] and col["STRAND"][7] == [0] and col["gnomAD_AF"][7] == [3.25] and col["MAX_AF"][7] == [4.0]
    assert col["Consequence"][8] == [b"x"] and col["SYMBOL"][8] == [None]                                              # C string: cut at the NUL
    assert all(v[9] == [None] for v in col.values())
    assert col["DISTANCE"][10] == [INT_MIN] and col["MOTIF_POS"][10] == [INT_MIN] and col["SpliceAI_pred_DS_AG"][10] == [0.25]
    assert all(v[11] is None and v[12] is None for v in col.values())                                                 # tag absent / zero-length value
    assert col["Allele"][13] == [b"A"] and col["Consequence"][13] == [b"B"]                                            # the value's bytes, whatever its BCF type


def test_edge_schemas():
    for tag in ("BCSQ", "ANN", "VEP", "vep"):
        t = orc.bcf_read(*_case("tag_" + tag))
        assert [c["name"] for c in t["cols"]][7:18] == ["VEP_" + f for f in vep_cases.FIELDS]
        assert orc.bcf_col_py(t["by_name"]["VEP_Allele"])[1] == [b"T", b"G"]
    t = orc.bcf_read(*_case("declared_integer"))                               # bcf_get_info_string: type mismatch -> no annotation
    assert all(v is None for v in orc.bcf_col_py(t["by_name"]["VEP_Allele"])) and t["n_rows"] == 3
    t = orc.bcf_read(*_case("no_format_in_description"))
    assert not [c for c in t["cols"] if c["name"].startswith("VEP_")]
    t = orc.bcf_read(*_case("one_field"))
    assert [c["name"] for c in t["cols"]][7:9] == ["VEP_Allele", "INFO_DP"]
    t = orc.bcf_read(*_case("format_to_end_of_value"))                         # unquoted Description: the list runs to the end of the value; a trailing '|' adds an empty name
    assert [c["name"] for c in t["cols"]][7:10] == ["VEP_A", "VEP_B_AF", "VEP_"]
    t = orc.bcf_read(*_case("csq_beats_ann"))
    assert [c["name"] for c in t["cols"]][7:9] == ["VEP_P", "VEP_Q"] and orc.bcf_col_py(t["by_name"]["VEP_Q"]) == [[b"csq2", b"d"]]


def test_edge_tidy_rows():
    t = orc.bcf_read(*_case("samples_tidy"))
    a = orc.bcf_col_py(t["by_name"]["VEP_Allele"])
    assert t["n_rows"] == 15
    assert a[0] == [b"T"] and a[1] is None and a[2] is None and a[3] == [b"T", b"G"] and a[4] is None    # bcf_reader.c:1370-1373
    w = orc.bcf_read(*_case("samples_wide"))
    assert orc.bcf_col_py(w["by_name"]["VEP_Allele"])[:2] == [[b"T"], [b"T", b"G"]]


# ---- GPU parity -------------------------------------------------------------------------------------------------------------
def _check(data, tidy=False, **kw):
    import duckhts_amd
    exp = orc.bcf_read(data, tidy)
    got = duckhts_amd.read_bcf(data, tidy=tidy, **kw)
    d = orc.bcf_cols_diff(exp, got)
    assert d is None, d
    return exp, got


@pytest.mark.gpu
def test_gpu_fixture(fixture):
    data, fields, rows, _ = fixture
    exp, got = _check(data)
    col = {n: orc.bcf_col_py(got["by_name"][n]) for n in ("VEP_Allele", "VEP_SYMBOL")}
    assert (col["VEP_Allele"][0][0], col["VEP_SYMBOL"][0][0]) == (b"T", b"WASH7P")   # duckhts.test:116-121
    _check(data, max_blocks=1)


@pytest.mark.gpu
@pytest.mark.parametrize("name,data,tidy", vep_cases.edge_cases(), ids=[c[0] for c in vep_cases.edge_cases()])
def test_gpu_edge_cases(name, data, tidy):
    _check(data, tidy)


def _long_vep(n, seed=4):
    """annotation strings of up to thousands of transcripts: empty pieces, short transcripts, white space, '.', numbers that do and do not parse"""
    import random
    rnd = random.Random(seed)
    fmt = "Allele|Consequence|SYMBOL|DISTANCE|STRAND|gnomAD_AF|HGVSc"
    hdr = ["##fileformat=VCFv4.2", "##contig=<ID=1,length=1000000>",
           '##INFO=<ID=CSQ,Number=.,Type=String,Description="Consequence annotations from Ensembl VEP. Format: %s">' % fmt, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    out = ["\n".join(hdr)]
    for r in range(n):
        ntr = rnd.choice([0, 1, 2, 40, 70, 130, 1100, 2300] if r % 9 == 0 else [1, 3, 20, 40])
        trs = []
        for _ in range(ntr):
            k = rnd.random()
            if k < 0.05:
                trs.append("")
            elif k < 0.1:
                trs.append("|".join(["A"] * rnd.randrange(1, 4)))
            else:
                trs.append("|".join([rnd.choice("ACGT"), rnd.choice(["missense_variant", " intron_variant ", ".", ""]), "GENE%d" % rnd.randrange(900),
                                     rnd.choice(["", "12", "-7", "+3", "1x", " 44 ", "99999999999999999999", "."]), rnd.choice(["1", "-1", ""]),
                                     rnd.choice(["", "0.25", "1e-5", "."]), "x" * rnd.choice([0, 1, 7, 8, 9, 30, 200])] + (["extra"] if k > 0.95 else [])))
        info = "CSQ=" + ",".join(trs) if (ntr or r % 2) else "."
        if r % 13 == 0:
            info = "CSQ=,,,"
        out.append("1\t%d\t.\tA\tC\t.\t.\t%s" % (100 + r, info))
    return ("\n".join(out) + "\n").encode()


@pytest.mark.gpu
@pytest.mark.parametrize("wave", ["0", "1"])
def test_gpu_vep_wave_per_row(wave, fixture, monkeypatch):
    """bcf_vep_wave (a wave per row and VEP column, a transcript per lane) and the lane-per-row form of bcf_cells, each forced onto the
    fixture, the edge cases and long annotation strings: the same columns as the oracle's"""
    import duckhts_amd
    monkeypatch.setenv("DHTS_VEP_WAVE", wave)
    _check(fixture[0])
    _check(fixture[0], max_blocks=1)
    for name, data, tidy in vep_cases.edge_cases():
        _check(data, tidy)
    txt = _long_vep(120)
    exp, got = _check(txt)
    assert max(len(x) for x in orc.bcf_col_py(got["by_name"]["VEP_SYMBOL"]) if x is not None) >= 1000
    names = [c["name"] for c in exp["cols"]]
    for want in (["VEP_DISTANCE"], ["VEP_HGVSc", "POS", "VEP_Allele"], ["VEP_gnomAD_AF", "VEP_STRAND"]):
        sub = duckhts_amd.read_bcf(txt, columns=[names.index(w) for w in want], max_blocks=1)
        assert orc.bcf_cols_diff({"n_rows": exp["n_rows"], "cols": [exp["by_name"][w] for w in want]}, sub) is None, want


@pytest.mark.gpu
def test_gpu_projection_of_vep_columns(fixture):
    import duckhts_amd
    data, fields, rows, exp = fixture
    names = [c["name"] for c in exp["cols"]]
    want = ["VEP_gnomAD_AF", "POS", "VEP_SYMBOL", "VEP_DISTANCE", "INFO_AF", "VEP_Allele"]
    got = duckhts_amd.read_bcf(data, columns=[names.index(w) for w in want])
    sub = {"n_rows": exp["n_rows"], "cols": [exp["by_name"][w] for w in want]}
    assert orc.bcf_cols_diff(sub, got) is None


@pytest.mark.gpu
def test_gpu_vep_through_the_table_function(fixture, tmp_path):
    """read_bcf over the DuckDB C API (tests/minihost): VEP_* LIST vectors with NULL elements, chunk for chunk against the oracle"""
    from test_duckdb_surface import compare_bcf
    data, fields, rows, _ = fixture
    compare_bcf(data, tmp_path)
    names = [c["name"] for c in orc.bcf_read(data)["cols"]]
    compare_bcf(data, tmp_path, proj=[names.index("VEP_SYMBOL"), names.index("VEP_gnomAD_AF"), 1, names.index("VEP_DISTANCE")])
    cs = {n: (d, t) for n, d, t in vep_cases.edge_cases()}
    for name in ("csq_values", "samples_tidy", "many", "declared_integer"):
        compare_bcf(cs[name][0], tmp_path, tidy=cs[name][1])
