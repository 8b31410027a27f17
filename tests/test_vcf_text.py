"""read_bcf on VCF TEXT (plain or BGZF), sites-only files: vcf_hdr_read + vcf_parse (htslib vcf.c:2594-2680, 3987-4165) in front of the
same column writers.

CPU: the oracle's text restatement pinned on the reference's own fixtures -- test/data/test_vep.vcf read as TEXT (duckhts.test:107-121
literally, and cell for cell against the independent BCF re-encoding of tests/vep_cases.py) and test/data/no_contig.vcf.gz (duckhts.test:
395-397: a contig the header does not define) -- plus hand-derived number / field / undefined-name cases.  GPU: the HIP path against the
oracle, bit for bit, through the C ABI and through the table function.
"""
import math
import os

import numpy as np
import pytest

import orc
import vcf_text_cases as V
import vep_cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = V.all_cases()


def _col(t, name):
    return orc.bcf_col_py(t["by_name"][name])


def test_reference_fixture_test_vep_vcf_as_text():
    txt = vep_cases.fixture_text().encode()
    t = orc.bcf_read(txt)
    assert t["n_rows"] == 802 and t["status"] == 0
    assert _col(t, "VEP_Allele")[0] is not None                                   # duckhts.test:108-113
    assert (_col(t, "VEP_Allele")[0][0], _col(t, "VEP_SYMBOL")[0][0]) == (b"T", b"WASH7P")   # duckhts.test:116-121
    bcf, fields, rows = vep_cases.fixture_bcf()                                   # the same records encoded to BCF2 by an independent Python reading of the text
    assert orc.bcf_cols_diff(t, orc.bcf_read(bcf)) is None
    assert [int(p) for p in _col(t, "POS")] == [p for p, _ in rows]


def test_reference_fixture_no_contig_vcf_gz():
    data = open(os.path.join(GOLD, "no_contig.vcf.gz"), "rb").read()
    t = orc.bcf_read(data)
    assert t["n_rows"] == 1 and t["status"] == 0                                  # duckhts.test:395-397
    assert (_col(t, "CHROM"), _col(t, "POS"), _col(t, "REF"), _col(t, "ALT"), _col(t, "QUAL"), _col(t, "FILTER"), _col(t, "INFO_DP")) == \
           ([b"chr1"], [1], [b"A"], [[b"T"]], [None], [[b"PASS"]], [1])


def test_numbers():
    t = orc.bcf_read(dict(CASES)["numbers_plain"])
    assert orc.bcf_cols_diff(t, orc.bcf_read(dict(CASES)["numbers_bgzf"])) is None
    f32 = lambda x: float(np.float32(x))
    q = _col(t, "QUAL")
    assert q[:5] == [30.0, None, 12.0, 100.0, 7.25] and q[5] == 0.0 and q[6] == math.inf and q[7] == f32(0.1) and q[8:] == [f32(3.98e-06), 0.5]      # atof: prefix parse, "-" -> 0
    assert _col(t, "INFO_DP") == [35, 5, None, None, None, None, None, None, None, None]       # "+5"; 99999999999 and "." and "" are missing
    ac = _col(t, "INFO_AC")
    assert ac[1] == [0, 0, 12, 7]                                                  # "-" and "+" convert to 0 (the end pointer moves past the sign), "12abc" to 12, "" is missing
    assert ac[2] == [2147483647, -2147483640]                                      # BCF_MIN_BT_INT32 = INT32_MIN + 8; everything outside is missing
    fv = _col(t, "INFO_FV")
    assert fv[1][:5] == [5.0, -0.0, 12.5, f32(1e-3), 100.0] and math.copysign(1, fv[1][1]) == -1
    assert math.isnan(fv[1][5]) and fv[1][6:] == [math.inf, -math.inf, 16.0]       # strtod's forms: nan, inf, hex; "abc", "" and "." are missing
    assert fv[3] == [f32(0.1234567890123), f32(0.12345678901234), f32(0.123456789012345), f32(1234567890123456789)]
    assert fv[4] == [1.5, -2.5, 4.0, 1.0, f32(1e-15), f32(1e14)]                    # leading white space is skipped, trailing garbage ignored
    assert fv[6] == [f32(3.4028235e38), math.inf, 0.0, 0.0] and fv[7][3] == 16777216.0
    mq = _col(t, "INFO_MQ")
    assert mq[:5] == [59.5, 0.5, None, f32(123456789012345.5), 3.5] and mq[5] is None and mq[8] == 1500.0
    assert _col(t, "INFO_DB") == [True, False, False, False, False, True, False, False, False, False]
    # exponent forms and the edges of the exact conversion (Clinger's fast path on the device, strtod on the host beyond it)
    assert fv[8] == [f32(3.98e-06), f32(1.23456e-05), f32(1e22), f32(1e23), f32(123456789012345e8), f32(1e-22), f32(1e-23), 0.0, math.inf, 1.0, 1.0, 100.0, 5.0, 0.0, 0.0, 1.0, f32(9.99999999999999e22)]
    assert fv[9] == [0.5, -0.25, 0.125, f32(1e-21), f32(1e21), f32(1e-22), f32(12345678901234e9)]


def test_fields_and_line_endings():
    t = orc.bcf_read(dict(CASES)["fields"])
    assert orc.bcf_cols_diff(t, orc.bcf_read(dict(CASES)["fields_crlf_no_final_eol"])) is None      # "\r\n" and a last line without newline
    assert t["n_rows"] == 19 and t["status"] == 0
    assert _col(t, "ID")[:2] == [b"rs1;rs2", None] and _col(t, "REF")[:2] == [b"ACGT", b"."]           # an empty REF prints as "." (bcf_fmt_array of 0 bytes)
    alt = _col(t, "ALT")
    assert alt[0] == [b"A", b"<DEL>", b"ACGTT"] and alt[1] == [b"."] and alt[2] == [] and alt[3] == [b"A", b"."] and alt[4] == [b".", b"A"]
    flt = _col(t, "FILTER")
    assert flt[5] == [b"PASS"] and flt[6] == [b"q10", b"s50"] and flt[7] == [b"q10"] and flt[9] == [b"DP"]
    dp = _col(t, "INFO_DP")
    assert dp[10:16] == [1, 2, 3, 4, 5, 6]                                           # ";;", a leading ';', an empty key, '=' inside a value, a trailing ';', a repeated key (first wins)
    assert _col(t, "INFO_ANN_S")[13] == b"a=b=c"
    assert _col(t, "POS")[16:] == [0, 30, 2147483646] and _col(t, "CHROM")[18] == b"chr2"


def test_undefined_names_get_dummy_definitions():
    t = orc.bcf_read(dict(CASES)["undefined_names"])
    assert t["n_rows"] == 4 and t["status"] == 0 and len(t["cols"]) == 16           # the columns were bound before the records were read
    assert _col(t, "CHROM") == [b"chrUn_1", b"chr2", b"chrUn_2", b"chrUn_1"]          # fix_chromosome (vcf.c:3744-3761)
    assert _col(t, "FILTER") == [[b"lowq"], [b"lowq", b"other"], [b"PASS"], [b"other"]]
    assert _col(t, "INFO_DP") == [1, None, None, None] and _col(t, "INFO_AF")[3] == [0.5]
    t3 = orc.bcf_read(dict(CASES)["undefined_names_same_tails"])
    assert t3["n_rows"] == 11 and t3["status"] == 0
    assert _col(t3, "CHROM") == [b"chrUn1", b"c8rUn1", b"2hrUn1", b"chrUn1", b"xbrUn", b"abrUn", b"chr_unplaced_1", b"chr_unplXced_1", b"Xhr_unplaced_1", b"c8rUn1", b"chr_unplaced_1"]
    assert _col(t3, "FILTER")[:3] == [[b"lowq7"], [b"Xowq7"], [b"lXwq7"]]
    t2 = orc.bcf_read(dict(CASES)["undefined_names_bgzf_small_blocks"])
    assert t2["n_rows"] == 160 and _col(t2, "CHROM")[:4] == _col(t, "CHROM")


@pytest.mark.parametrize("name,rows", [("bad_pos", 3), ("bad_pos_overflow", 3), ("pos_beyond_62_bits", 3), ("too_few_columns", 3), ("empty_line", 3),
                                       ("undefined_contig_with_bad_name", 3), ("nul_in_line", 2)])
def test_the_first_bad_line_ends_the_scan(name, rows):
    t = orc.bcf_read(dict(CASES)[name])
    assert t["n_rows"] == rows and t["status"] < 0                                  # rows before it are kept, silently (bcf_reader.c:1319-1349)


def test_reference_fixture_formatcols_vcf_gz():
    t = orc.bcf_read(open(os.path.join(GOLD, "formatcols.vcf.gz"), "rb").read())
    assert t["n_rows"] == 1 and t["status"] == 0                                  # duckhts.test:16-18
    assert (_col(t, "CHROM"), _col(t, "POS"), _col(t, "ID"), _col(t, "REF")) == ([b"1"], [100], [b"a"], [b"A"])      # duckhts.test:22-24
    assert [c["name"] for c in t["cols"]][7:] == ["FORMAT_S_S1", "FORMAT_S_S\u00b2", "FORMAT_S_S3"]
    assert [_col(t, n)[0] for n in ("FORMAT_S_S1", "FORMAT_S_S\u00b2", "FORMAT_S_S3")] == [b"a", b"bbbbbbb", b"ccccccccc"]


@pytest.mark.parametrize("tidy", [False, True])
def test_upstream_text_form_of_vcf_file_bcf_reads_like_the_binary(tidy):
    """htslib's test/tabix/vcf_file.vcf is the text the reference's vcf_file.bcf was made from: both readers must give one table
    (23 columns x 15 rows wide, 30 rows tidy; INFO of every type, FORMAT GT / GQ / DP / GL / TT of two samples)"""
    a = orc.bcf_read(open(os.path.join(GOLD, "vcf_file.vcf"), "rb").read(), tidy)
    b = orc.bcf_read(open(os.path.join(GOLD, "vcf_file.bcf"), "rb").read(), tidy)
    assert a["n_rows"] == (30 if tidy else 15) and orc.bcf_cols_diff(a, b) is None


def test_sample_columns():
    t = orc.bcf_read(dict(CASES)["samples"])
    assert t["n_rows"] == 14 and t["status"] == 0
    gt = [_col(t, "FORMAT_GT_S%d" % k) for k in (1, 2, 3)]
    assert [g[0] for g in gt] == [b"0/1", b"1|1", b"./."] and [g[1] for g in gt] == [b"0|1|2", b".", b"1"] and [g[3] for g in gt] == [b"0/1", None, b"1/1"]
    assert _col(t, "FORMAT_GQ_S2")[3] == 6 and _col(t, "FORMAT_GQ_S2")[7] == 0       # a sample that leaves trailing fields out; "-" converts to 0
    assert _col(t, "FORMAT_GL_S1")[6] == [0.0] and _col(t, "FORMAT_GL_S3")[6] == []   # an empty Float stores what strtod returned; ".,." is two missing values
    assert _col(t, "FORMAT_GL_S1")[4] == [0.5, float(np.float32(1e-3)), 5.0] and math.isnan(_col(t, "FORMAT_GL_S2")[4][0])
    assert all(_col(t, "FORMAT_GT_S1")[r] is None for r in (8, 9))                    # FORMAT "." and a line without FORMAT
    assert _col(t, "FORMAT_GT_S1")[11] == b"0/1"                                      # duplicate tag: the first occurrence is kept
    assert len(_col(t, "FORMAT_AD_S1")[13]) == 17
    v44 = orc.bcf_read(dict(CASES)["samples_v44_bgzf"])
    assert [_col(v44, "FORMAT_GT_S%d" % k)[0] for k in (1, 2, 3)] == [b"0/1", b"1|0", b"0"]   # VCFv4.4: a leading '/' or '|' is a phasing prefix


@pytest.mark.parametrize("name", sorted(V.SAMPLE_ERRORS))
def test_sample_column_errors_end_the_scan(name):
    t = orc.bcf_read(dict(CASES)["samples_" + name])
    assert t["n_rows"] == 2 and t["status"] < 0


# ---- GPU parity -------------------------------------------------------------------------------------------------------------
def _check(data, tidy=False, **kw):
    import duckhts_amd
    exp = orc.bcf_read(data, tidy)
    got = duckhts_amd.read_bcf(data, tidy=tidy, **kw)
    d = orc.bcf_cols_diff(exp, got)
    assert d is None, d
    assert (got["status"] == 1) == (exp["status"] == 0), (got["status"], exp["status"])
    return exp, got


@pytest.mark.gpu
@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_gpu_cases(name, data):
    _check(data)
    if len(data) > 3000:
        _check(data, max_blocks=1)


def _long_info_lines(n, seed=3):
    """gnomAD-shaped sites-only lines: hundreds of numeric INFO keys, flags, a long vep string, END= / SVLEN= on some, a few odd fields"""
    import random
    rnd = random.Random(seed)
    keys = [("K%03d" % i, rnd.choice(["Integer", "Float", "String", "Flag"]), rnd.choice(["1", "A", "."])) for i in range(260)]
    hdr = ["##fileformat=VCFv4.2", '##FILTER=<ID=RF,Description="x">', "##contig=<ID=22,length=50818468>", '##ALT=<ID=DEL,Description="d">',
           '##INFO=<ID=END,Number=1,Type=Integer,Description="e">', '##INFO=<ID=SVLEN,Number=.,Type=Integer,Description="s">',
           '##INFO=<ID=vep,Number=.,Type=String,Description="Consequence annotations from Ensembl VEP. Format: Allele|Consequence|SYMBOL">']
    hdr += ['##INFO=<ID=%s,Number=%s,Type=%s,Description="x">' % (k, "0" if t == "Flag" else num, t) for k, t, num in keys]
    hdr.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO")
    out = ["\n".join(hdr)]
    pos = 1000
    for r in range(n):
        pos += rnd.randrange(1, 500)
        info = []
        for k, t, num in keys:
            if rnd.random() < 0.15:
                continue
            if t == "Flag":
                info.append(k)
            elif t == "Integer":
                info.append("%s=%s" % (k, ",".join(rnd.choice([str(rnd.randrange(-9, 250000)), ".", "+7", "99999999999", "1x"]) for _ in range(rnd.choice([1, 1, 1, 3])))))
            elif t == "Float":
                info.append("%s=%s" % (k, ",".join(rnd.choice(["%.5e" % rnd.random(), "%.3f" % (rnd.random() * 100), ".", "1e-30", "0.12345678901234567", "nan", "-inf"]) for _ in range(rnd.choice([1, 1, 2])))))
            else:
                info.append("%s=%s" % (k, "s" * rnd.choice([0, 1, 5, 95, 96, 97, 300])))
        info.append("vep=" + ",".join("|".join(["A", "missense_variant", "GENE%d" % rnd.randrange(900)] + ["x" * rnd.randrange(20)]) for _ in range(rnd.randrange(1, 120))))
        if r % 7 == 0:
            info.insert(rnd.randrange(len(info)), "END=%d" % (pos + 400))
        if r % 11 == 0:
            info.insert(rnd.randrange(len(info)), "SVLEN=-300")
        if r % 13 == 0:
            info.insert(rnd.randrange(len(info)), "UNDEFINED_%d=4" % (r % 3))
        if r % 17 == 0:
            info.insert(rnd.randrange(len(info)), "")                    # ";;"
        if r % 19 == 0:
            info.insert(rnd.randrange(len(info)), "=5")
        alt = "<DEL>" if r % 11 == 0 else rnd.choice("ACGT")
        out.append("22\t%d\t.\tA\t%s\t%s\t%s\t%s%s" % (pos, alt, rnd.choice([".", "50", "1e-30"]), rnd.choice([".", "PASS", "RF"]), ";".join(info), ";" if r % 23 == 0 else ""))
    return ("\n".join(out) + "\n").encode()


@pytest.mark.gpu
def test_gpu_wave_per_line_encoder(monkeypatch):
    """vcf_encode_wave (a wave per line, INFO fields one per lane) against the oracle: forced onto every case of the lane-per-line tests, and on
    long INFO lines"""
    monkeypatch.setenv("DHTS_VCF_WAVE", "1")
    for name, data in CASES:
        _check(data)
    txt = _long_info_lines(300)
    exp, got = _check(txt)
    assert got["n_rows"] == 300
    _check(txt, max_blocks=1)
    import bamwriter
    _check(bamwriter.bgzf_file(txt, payload=30000), max_blocks=2)
    _check(vep_cases.fixture_text().encode())


def _many_host_numbers(n_info, n_fmt):
    """more numbers outside the device's conversion path than one batch used to have room for: 17-digit INFO floats (Python's repr), tiny
    FORMAT p-values (1e-30), a few inf / nan"""
    import random
    rnd = random.Random(11)
    hdr = ["##fileformat=VCFv4.2", "##contig=<ID=1,length=248956422>", '##INFO=<ID=P,Number=.,Type=Float,Description="p">',
           '##FORMAT=<ID=PV,Number=1,Type=Float,Description="pv">', "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\tS3\tS4"]
    lines = ["\n".join(hdr)]
    per = 64
    for r in range(max(n_info // per, n_fmt // 4) + 1):
        vals = ",".join(repr(rnd.random() * 10 ** rnd.randrange(-3, 4)) if k % 31 else rnd.choice(["inf", "nan", "1e-300", "0x1p-3"]) for k in range(per))
        smp = "\t".join(rnd.choice(["1e-30", "3.5e-45", "1e-30", "0.5"]) for _ in range(4))
        lines.append("1\t%d\t.\tA\tC\t%s\t.\tP=%s\tPV\t%s" % (100 + r, rnd.choice([".", "1e-40", "30"]), vals, smp))
    return ("\n".join(lines) + "\n").encode()


@pytest.mark.gpu
@pytest.mark.parametrize("wave", ["0", "1"])
def test_gpu_more_host_converted_numbers_than_the_initial_room(wave, monkeypatch):
    """the per-batch records of tokens for the host (undefined names / strtod checks, numbers to convert) grow and the pass is repeated; the
    tokens travel in one gather + one copy.  Small initial room, so that a few thousand numbers already overflow it"""
    monkeypatch.setenv("DHTS_VCF_CAPS", "257")
    monkeypatch.setenv("DHTS_VCF_WAVE", wave)
    data = _many_host_numbers(20000, 3000)
    exp, got = _check(data)
    assert got["n_rows"] == exp["n_rows"] > 300
    _check(data, max_blocks=1)


@pytest.mark.gpu
def test_gpu_a_million_host_converted_numbers_in_one_batch():
    """the sizes the advisor named: > 1M 17-digit INFO floats and > 65536 FORMAT floats such as 1e-30 in one batch, default room"""
    data = _many_host_numbers(1_100_000, 70_000)
    import duckhts_amd
    got = duckhts_amd.read_bcf(data)
    n = data.count(b"\n") - 5
    assert got["n_rows"] == n and got["status"] == 1
    p = orc.bcf_col_py(got["by_name"]["INFO_P"])
    assert len(p) == n and all(len(x) == 64 for x in p[:50])
    # spot check against Python's own float(): f32 of the 17-digit text
    import numpy as np
    first = data.split(b"\n")[5].split(b"\t")[7][2:].split(b",")
    want = np.array([float.fromhex(t.decode()) if t.startswith(b"0x") else float(t) for t in first], np.float64).astype(np.float32)
    have = np.array(p[0], np.float32)
    assert np.array_equal(np.isnan(want), np.isnan(have)) and np.array_equal(want[~np.isnan(want)], have[~np.isnan(have)])


@pytest.mark.gpu
def test_gpu_long_lines_choose_the_wave_encoder_by_themselves():
    txt = _long_info_lines(200, seed=5)
    exp, got = _check(txt)
    assert got["n_rows"] == 200


@pytest.mark.gpu
def test_gpu_reference_fixtures():
    txt = vep_cases.fixture_text().encode()
    exp, got = _check(txt)
    assert got["n_rows"] == 802
    assert (orc.bcf_col_py(got["by_name"]["VEP_Allele"])[0][0], orc.bcf_col_py(got["by_name"]["VEP_SYMBOL"])[0][0]) == (b"T", b"WASH7P")   # duckhts.test:116-121
    _check(txt, max_blocks=2)
    import bamwriter
    _check(bamwriter.bgzf_file(txt, payload=5000), max_blocks=3)
    exp, got = _check(open(os.path.join(GOLD, "no_contig.vcf.gz"), "rb").read())
    assert got["n_rows"] == 1 and orc.bcf_col_py(got["by_name"]["CHROM"]) == [b"chr1"]   # duckhts.test:395-397
    exp, got = _check(open(os.path.join(GOLD, "formatcols.vcf.gz"), "rb").read())
    assert got["n_rows"] == 1 and orc.bcf_col_py(got["by_name"]["ID"]) == [b"a"]          # duckhts.test:16-24
    for tidy in (False, True):
        exp, got = _check(open(os.path.join(GOLD, "vcf_file.vcf"), "rb").read(), tidy=tidy)
        import duckhts_amd
        assert orc.bcf_cols_diff(got, duckhts_amd.read_bcf(open(os.path.join(GOLD, "vcf_file.bcf"), "rb").read(), tidy=tidy)) is None


@pytest.mark.gpu
def test_gpu_projection_and_sample_columns():
    import duckhts_amd
    data = dict(CASES)["many_bgzf"]
    exp = orc.bcf_read(data)
    names = [c["name"] for c in exp["cols"]]
    want = ["INFO_FV", "POS", "QUAL", "INFO_AF"]
    got = duckhts_amd.read_bcf(data, columns=[names.index(w) for w in want])
    assert orc.bcf_cols_diff({"n_rows": exp["n_rows"], "cols": [exp["by_name"][w] for w in want]}, got) is None


@pytest.mark.gpu
@pytest.mark.parametrize("wave", ["0", "1"])
def test_gpu_projection_reaches_the_text_encoder(wave, monkeypatch):
    """the INFO keys the projected columns read are all the encoder writes into the records it makes (VcfArgs::info_keep): every projection
    gives the columns of the full read -- a single key, the annotation tag behind a VEP_ column, no INFO column at all (rows, POS, and the
    END= / SVLEN= interval behind a region filter), random subsets --, names without a definition are still reported, the 65535-entry
    limit still counts every field"""
    import random
    import duckhts_amd
    monkeypatch.setenv("DHTS_VCF_WAVE", wave)
    txt = _long_info_lines(120, seed=8)
    exp = orc.bcf_read(txt)
    names = [c["name"] for c in exp["cols"]]
    rnd = random.Random(5)
    info_cols = [n for n in names if n.startswith("INFO_")]
    projections = [["POS"], ["CHROM", "POS", "REF", "ALT", "QUAL", "FILTER"], ["INFO_END"], ["VEP_SYMBOL"], ["INFO_vep", "POS"], ["VEP_Consequence", "INFO_SVLEN", "ID"]]
    projections += [rnd.sample(info_cols, k) + ["POS"] for k in (1, 3, 40)] + [info_cols]
    for want in projections:
        for kw in ({}, {"max_blocks": 1}):
            got = duckhts_amd.read_bcf(txt, columns=[names.index(w) for w in want], **kw)
            d = orc.bcf_cols_diff({"n_rows": exp["n_rows"], "cols": [exp["by_name"][w] for w in want]}, got)
            assert d is None, (want[:4], d)
    # a region filter on a projection without INFO columns: the interval comes from END= / SVLEN= all the same
    import bamwriter
    from test_vcf_region import build_index
    z = bamwriter.bgzf_file(txt, payload=20000)
    raw, tbi = build_index(z, 0)                                         # (the index build projects nothing)
    full = duckhts_amd.read_bcf(z, region="22:3000-9000", index=tbi)
    part = duckhts_amd.read_bcf(z, region="22:3000-9000", index=tbi, columns=[names.index("POS")])
    assert part["n_rows"] == full["n_rows"] > 0
    monkeypatch.setenv("DHTS_VCF_KEEP", "0")                             # every key encoded whatever is projected: the same index
    assert build_index(z, 0)[0] == raw
    monkeypatch.delenv("DHTS_VCF_KEEP")
    assert orc.bcf_cols_diff({"n_rows": full["n_rows"], "cols": [c for c in full["cols"] if c["name"] == "POS"]}, part) is None
    # 65535 INFO entries are too many whatever is projected (vcf_parse_info counts every field)
    hdr = "##fileformat=VCFv4.2\n##contig=<ID=1>\n##INFO=<ID=A,Number=1,Type=Integer,Description=\"x\">\n##INFO=<ID=B,Number=0,Type=Flag,Description=\"x\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    many = (hdr + "1\t5\t.\tA\tC\t.\t.\tA=1\n1\t6\t.\tA\tC\t.\t.\t" + ";".join(["B"] * 65536) + "\n1\t7\t.\tA\tC\t.\t.\tA=2\n").encode()
    e2 = orc.bcf_read(many)
    n2 = [c["name"] for c in e2["cols"]]
    for want in (["POS"], ["INFO_A"], n2):
        g2 = duckhts_amd.read_bcf(many, columns=[n2.index(w) for w in want])
        assert g2["n_rows"] == e2["n_rows"] and (g2["status"] == 1) == (e2["status"] == 0), (want[:3], g2["n_rows"], e2["n_rows"], g2["status"], e2["status"])


@pytest.mark.gpu
@pytest.mark.parametrize("wave", ["0", "1"])
def test_gpu_projection_without_format_columns_still_validates_the_samples(wave, monkeypatch):
    """no FORMAT column projected, or some: the encoder leaves the sample data (the other keys' values) out of the records it makes (VcfArgs::fmt_none / fmt_keep), but the measure pass has
    walked every sample column -- a line whose samples do not parse ends the scan whatever is projected -- and tidy mode still has a row per
    record and sample"""
    import duckhts_amd
    monkeypatch.setenv("DHTS_VCF_WAVE", wave)
    names_of = lambda t: [c["name"] for c in t["cols"]]
    for name, data in CASES:
        if not name.startswith("samples"):
            continue
        for tidy in (False, True):
            exp = orc.bcf_read(data, tidy)
            if not exp["cols"]:
                continue
            names = names_of(exp)
            fmt = [n for n in names if n.startswith("FORMAT_")]
            some_fmt = [[fmt[0]], [fmt[-1], "POS"], fmt[1:4:2] + ["ID"]] if len(fmt) >= 4 else ([[fmt[0]]] if fmt else [])     # a subset of the FORMAT keys: the others are validated, not written (VcfArgs::fmt_keep)
            for want in [["POS"], ["CHROM", "POS", "REF", "ALT", "QUAL", "FILTER"] + [n for n in names if n.startswith("INFO_")][:2], ["SAMPLE_ID", "POS"] if "SAMPLE_ID" in names else ["ID"]] + some_fmt:
                got = duckhts_amd.read_bcf(data, tidy=tidy, columns=[names.index(w) for w in want], max_blocks=1 if len(data) > 3000 else 0)
                d = orc.bcf_cols_diff({"n_rows": exp["n_rows"], "cols": [exp["by_name"][w] for w in want]}, got)
                assert d is None, (name, tidy, want, d)
                assert (got["status"] == 1) == (exp["status"] == 0), (name, tidy, want, got["status"], exp["status"])


def _many_format_keys(n_keys, n_smp=3, n_lines=5, declared=True):
    import random
    rnd = random.Random(n_keys)
    hdr = ["##fileformat=VCFv4.2", "##contig=<ID=1>", '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">']
    keys = ["GT"] + ["F%03d" % i for i in range(n_keys - 1)]
    types = {}
    for k in keys[1:]:
        types[k] = rnd.choice(["Integer", "Float", "String"])
        if declared or rnd.random() < 0.5:
            hdr.append('##FORMAT=<ID=%s,Number=%s,Type=%s,Description="x">' % (k, rnd.choice(["1", "."]), types[k]))
    hdr.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("S%d" % i for i in range(n_smp)))
    out = ["\n".join(hdr)]
    for r in range(n_lines):
        cols = []
        for _ in range(n_smp):
            vals = [rnd.choice(["0/1", "1|1", "./.", "0"])]
            for k in keys[1:rnd.randrange(1, len(keys) + 1)]:
                vals.append({"Integer": rnd.choice(["7", "1,2,3", ".", "-5"]), "Float": rnd.choice(["0.5", "1e-3,2", ".", "nan"]), "String": rnd.choice(["ab", "", "x,y", "."])}[types[k]])
            cols.append(":".join(vals))
        out.append("1\t%d\t.\tA\tC\t.\t.\t.\t%s\t%s" % (10 + r, ":".join(keys), "\t".join(cols)))
    return ("\n".join(out) + "\n").encode()


@pytest.mark.gpu
@pytest.mark.parametrize("wave", [None, "0", "1"])
def test_gpu_up_to_255_format_keys(wave, monkeypatch):
    """htslib keeps 255 FORMAT identifiers per line (MAX_N_FMT, vcf.c:3134); the lane-per-line encoder keeps tables for 32 and hands a batch with
    more to the wave encoder, whose tables live in LDS; the 256th key is an error in both"""
    if wave is not None:
        monkeypatch.setenv("DHTS_VCF_WAVE", wave)
    for n_keys in (31, 32, 33, 40, 255):
        exp, got = _check(_many_format_keys(n_keys))
        assert exp["status"] == 0 and exp["n_rows"] == 5, n_keys
    _check(_many_format_keys(60, declared=False))                         # names without a definition among them
    exp, got = _check(_many_format_keys(256))
    assert exp["status"] < 0 and exp["n_rows"] == 0
    mixed = _many_format_keys(20, n_lines=4) + b"\n".join(_many_format_keys(90, n_lines=3).split(b"\n")[-4:])   # (same samples; the keys of the second file are not declared in the first)
    _check(mixed)


@pytest.mark.gpu
def test_gpu_vcf_text_through_the_table_function(tmp_path):
    """the reference's own queries on its own files: read_bcf('test_vep.vcf') (plain text) and read_bcf('no_contig.vcf.gz')"""
    from test_duckdb_surface import compare_bcf, run_host
    compare_bcf(vep_cases.fixture_text().encode(), tmp_path)
    compare_bcf(open(os.path.join(GOLD, "no_contig.vcf.gz"), "rb").read(), tmp_path)
    compare_bcf(dict(CASES)["undefined_names_bgzf_small_blocks"], tmp_path)
    compare_bcf(dict(CASES)["many_plain"], tmp_path, proj=[0, 1, 5, 6, 8, 15])
    compare_bcf(open(os.path.join(GOLD, "formatcols.vcf.gz"), "rb").read(), tmp_path)
    compare_bcf(open(os.path.join(GOLD, "vcf_file.vcf"), "rb").read(), tmp_path)
    compare_bcf(dict(CASES)["samples_many_bgzf"], tmp_path, tidy=True)


@pytest.mark.gpu
def test_gpu_region_on_vcf_text_through_the_table_function(tmp_path):
    """duckhts.test:399-403: a region naming a contig nobody knows yields no iterator, i.e. zero rows; without an index the reference's
    message; a region the index knows is served (more in tests/test_vcf_region.py)"""
    import shutil
    from test_duckdb_surface import run_host
    fn = os.path.join(str(tmp_path), "no_contig.vcf.gz")
    shutil.copy(os.path.join(GOLD, "no_contig.vcf.gz"), fn)
    rc, out, _ = run_host(fn, named=[("region", "no_such_contig:1-10")], fn="read_bcf")
    assert rc == 3 and out == "ERROR init: Region query requires an index file (.tbi or .csi). Region: no_such_contig:1-10"   # bcf_reader.c:922-923
    shutil.copy(os.path.join(GOLD, "no_contig.vcf.gz.tbi"), fn + ".tbi")
    rc, out, _ = run_host(fn, named=[("region", "no_such_contig:1-10")], fn="read_bcf")
    assert rc == 0 and "rows=0 " in out                                              # duckhts.test:401-403
    rc, out, _ = run_host(fn, fn="read_bcf")
    assert rc == 0 and "rows=1 " in out                                              # duckhts.test:395-397
    rc, out, _ = run_host(fn, named=[("region", "chr1:1-1000")], fn="read_bcf")      # chr1 has no ##contig line: the tabix index names it
    assert rc == 0 and "rows=1 " in out
    rc, out, _ = run_host(fn, named=[("region", "chr1:2-3,no_such_contig")], fn="read_bcf")
    assert rc == 0 and "rows=0 " in out


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["many_bgzf", "many_plain", "samples_many_bgzf", "undefined_names_bgzf_small_blocks"])
def test_gpu_block_range_shards_of_text_concatenate(name):
    """block-range shards on VCF text: a shard that starts inside the file owns the lines that START in its blocks (lines synchronise on the
    newline; the line that runs out of a shard is finished from the next shard's blocks) -- every way of cutting reproduces the whole scan"""
    import duckhts_amd
    data = dict(CASES)[name]
    exp = orc.bcf_read(data)
    ctx = duckhts_amd.Context(0)
    ctx.open(data)
    nb = ctx.bgzf_index()
    ctx.close()
    if nb < 3:
        pytest.skip("a single block")
    for ways in (2, 3, min(7, nb), nb):
        cuts = [nb * k // ways for k in range(ways + 1)]
        spans, parts = [], []
        for r in range(ways):
            got = duckhts_amd.read_bcf(data, block_range=(cuts[r], cuts[r + 1], r > 0), max_blocks=(0, 1, 2)[r % 3])
            if exp["status"] == 0:
                assert got["status"] == 1, (name, ways, r, got["status"])
            spans.append((got["first_rec_uoff"], got["end_uoff"], got["n_rows"]))
            parts.append(got)
        if exp["status"] == 0:
            assert sum(p["n_rows"] for p in parts) == exp["n_rows"], (name, ways, [p["n_rows"] for p in parts])
            for col in ("POS", "REF"):
                key = "fixed" if col == "POS" else "sbytes"
                assert np.array_equal(np.concatenate([p["by_name"][col][key] for p in parts if p["n_rows"]]), exp["by_name"][col][key]), (name, ways, col)
            nonempty = [s for s in spans if s[2]]
            assert duckhts_amd.check_handoff(nonempty) == exp["n_rows"]
