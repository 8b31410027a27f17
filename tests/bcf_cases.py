"""Edge-case BCF inputs shared by the oracle tests (CPU) and the GPU parity tests."""
import random
import struct

import bcfwriter as W
from bcfwriter import END, MISSING

HDR_LINES = [
    '##FILTER=<ID=q10,Description="Quality below 10">',
    '##FILTER=<ID=s50,Description="Less than 50% of samples have data">',
    '##INFO=<ID=DP,Number=1,Type=Integer,Description="d">',
    '##INFO=<ID=AF,Number=A,Type=Float,Description="d">',
    '##INFO=<ID=AC,Number=A,Type=Integer,Description="d">',
    '##INFO=<ID=AN,Number=1,Type=Integer,Description="d">',
    '##INFO=<ID=MQ,Number=1,Type=Float,Description="d">',
    '##INFO=<ID=DB,Number=0,Type=Flag,Description="d">',
    '##INFO=<ID=SB,Number=4,Type=Integer,Description="d">',
    '##INFO=<ID=ANN_S,Number=1,Type=String,Description="d">',
    '##INFO=<ID=TAGS,Number=.,Type=String,Description="d">',
    '##INFO=<ID=VALS,Number=.,Type=Integer,Description="d">',
    '##INFO=<ID=FV,Number=.,Type=Float,Description="d">',
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="d">',
    '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="d">',
    '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
    '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="d">',
    '##FORMAT=<ID=PL,Number=G,Type=Integer,Description="d">',
    '##FORMAT=<ID=GL,Number=G,Type=Float,Description="d">',
    '##FORMAT=<ID=FT,Number=1,Type=String,Description="d">',
    '##FORMAT=<ID=HQ,Number=2,Type=Integer,Description="d">',
    '##FORMAT=<ID=SC,Number=1,Type=Float,Description="d">',
]
# dictionary ids given HDR_LINES (PASS = 0, then order of first appearance)
K = {"PASS": 0, "q10": 1, "s50": 2, "DP": 3, "AF": 4, "AC": 5, "AN": 6, "MQ": 7, "DB": 8, "SB": 9, "ANN_S": 10, "TAGS": 11, "VALS": 12,
     "FV": 13, "GT": 14, "GQ": 15, "AD": 16, "PL": 17, "GL": 18, "FT": 19, "HQ": 20, "SC": 21}
SAMPLES = ("S1", "S2", "S3")


def std_header(samples=SAMPLES, fileformat="VCFv4.2", extra=()):
    return W.header(HDR_LINES + list(extra), samples=samples, contigs=("chr1", "chr2", "chrX"), fileformat=fileformat)


def basic_records():
    ns = len(SAMPLES)
    r = []
    r.append(W.record(0, 99, 1, 59.2, b"", (b"C", b"T"), None,
                      [(K["DP"], W.tv_ints([35])), (K["AF"], W.tv_floats([0.5])), (K["AC"], W.tv_ints([2])), (K["AN"], W.tv_ints([4])),
                       (K["MQ"], W.tv_floats([59.5])), (K["DB"], b"\x00"), (K["SB"], W.tv_ints([1, 2, 3, 4])), (K["ANN_S"], W.tv_str(b"missense"))],
                      [W.fmt_ints(K["GT"], [W.gt(0, 1), W.gt(1, 1), W.gt(0, 0)]), W.fmt_ints(K["GQ"], [[99], [45], [MISSING]]),
                       W.fmt_ints(K["DP"], [[30], [300], [70000]]), W.fmt_ints(K["AD"], [[10, 20], [0, 300], [MISSING, MISSING]]),
                       W.fmt_ints(K["PL"], [[0, 30, 300], [255, 0, 1000], [0, 0, 0]]), W.fmt_floats(K["GL"], [[-0.1, -2.5, -30.0], [MISSING], [-1.0, -2.0]]),
                       W.fmt_strs(K["FT"], [b"PASS", b".", b"lowq;x"])], ns))
    r.append(W.record(0, 100, 4, None, b"rs1;rs2", (b"CAAA", b"C", b"CA", b""), [K["q10"], K["s50"]],
                      [(K["DP"], W.tv_ints([1000])), (K["AF"], W.tv_floats([0.25, MISSING, 1e-30])), (K["AC"], W.tv_ints([1, MISSING, 100000])),
                       (K["VALS"], W.tv_ints([MISSING, 5, END, 7])), (K["FV"], W.tv_floats([MISSING, MISSING])), (K["TAGS"], W.tv_str(b"a,b,,c,"))],
                      [W.fmt_ints(K["GT"], [W.gt(0, 1, phased=True), W.gt(2), W.gt(None, None)]), W.fmt_ints(K["HQ"], [[10, 20], [MISSING, 5], [END]])], ns))
    r.append(W.record(1, 0xFFFFFFFF - (1 << 32), 1, 0.0, b".", (b"N",), [K["PASS"]], [(K["DB"], W.tv_ints([1])), (K["TAGS"], W.tv_str(b"."))], [], ns))
    r.append(W.record(2, 5, 1, -0.0, b"\0x", (b"A", b"<DEL>"), None, [(K["ANN_S"], W.tv_str(b"ab\0cd")), (K["TAGS"], W.tv_str(b"\0")), (K["DP"], W.tv_ints([MISSING]))],
                      [W.fmt_ints(K["GT"], [W.gt(0, 1, 2), W.gt(1, 1) + [END], W.gt(70, 80, phased=True) + [END]]), W.fmt_floats(K["SC"], [[1.5], [MISSING], [END]])], ns))
    return r


def mismatch_records():
    """stored type differs from the header type (getter semantics vcf.c:6096-6131, 6221-6244)."""
    ns = len(SAMPLES)
    return [
        W.record(0, 1, 1, 1.0, b"m1", (b"A", b"C"), None,
                 [(K["DP"], W.tv_floats([1.5])), (K["MQ"], W.tv_ints([7])), (K["ANN_S"], W.tv_ints([65, 66, 0, 67], width=2)), (K["AN"], W.tv_str(b"x")),
                  (K["AC"], W.tv_floats([MISSING, 2.0, END])), (K["AF"], W.tv_ints([MISSING, 3, END])), (K["DB"], W.tv_str(b"yes")), (K["VALS"], b"\x00")],
                 [W.fmt_floats(K["GQ"], [[1.0], [MISSING], [END]]), W.fmt_ints(K["SC"], [[5], [MISSING], [END]]), W.fmt_floats(K["AD"], [[1.0, MISSING], [END, END], [2.0, 3.0]]),
                  W.fmt_ints(K["GL"], [[1, MISSING], [END, END], [2, 3]])], ns),
        # duplicate keys: the first occurrence wins (vcf.c:6064-6066); first value vector_end => NULL; all-missing list => empty list
        W.record(0, 2, 1, 2.0, b"m2", (b"A", b"C"), None,
                 [(K["DP"], W.tv_ints([1])), (K["DP"], W.tv_ints([2])), (K["VALS"], W.tv_ints([END, 4])), (K["AC"], W.tv_ints([MISSING, MISSING])), (K["AN"], W.tv_ints([END]))],
                 [W.fmt_ints(K["GQ"], [[1], [2], [3]]), W.fmt_ints(K["GQ"], [[4], [5], [6]]), W.fmt_ints(K["AD"], [[MISSING, MISSING], [END], [MISSING, 1]])], ns),
    ]


def qual_records():
    bits = [0x7F800002, 0x7FC00000, 0x7F800000, 0xFF800000, 0x00000001, 0x80000000, 0x7F800001, 0x3DCCCCCD, 0x7F7FFFFF, 0x00800000, 0x007FFFFF, 0xFFC12345]
    return [W.record(0, i, 1, alleles=(b"A",), qual_bits=b) for i, b in enumerate(bits)]


def gt_records():
    ns = len(SAMPLES)
    return [
        W.record(0, 1, 1, 1.0, b"g1", (b"A", b"C"), None, [], [W.fmt_ints(K["GT"], [W.gt(0), W.gt(1), W.gt(None)])], ns),
        W.record(0, 2, 1, 1.0, b"g2", (b"A", b"C"), None, [], [W.fmt_ints(K["GT"], [W.gt(0, 1, 1, phased=True), W.gt(0, 1) + [END], [END, END, END]])], ns),
        W.record(0, 3, 1, 1.0, b"g3", (b"A", b"C"), None, [], [W.fmt_ints(K["GT"], [W.gt(0, 200), W.gt(300, 1, phased=True), W.gt(None, 5)], width=2)], ns),
        W.record(0, 4, 1, 1.0, b"g4", (b"A", b"C"), None, [], [W.fmt_ints(K["GT"], [[MISSING, 4], [4, MISSING], [2, 5]], width=1)], ns),
        W.record(0, 5, 1, 1.0, b"g5", (b"A", b"C"), None, [], [W.fmt_ints(K["GT"], [W.gt(0, 1), W.gt(1, 0, phased=True), W.gt(100000, 1)], width=4)], ns),
        W.record(0, 6, 1, 1.0, b"g6", (b"A", b"C"), None, [], [W.fmt_ints(K["GQ"], [[1], [2], [3]])], ns),     # GT absent
    ]


def fuzz_records(seed, n, ns, n_ctg=3):
    rnd = random.Random(seed)
    out = []
    pos = 0

    def ints(k, lo=-200, hi=70000):
        vals = []
        for _ in range(k):
            x = rnd.random()
            vals.append(MISSING if x < 0.05 else rnd.randint(-100, 100) if x < 0.6 else rnd.randint(lo, hi))
        cut = rnd.randint(0, k) if rnd.random() < 0.15 else k
        return vals[:cut] + [END] * (k - cut) if rnd.random() < 0.5 else vals[:max(cut, 1)]

    def floats(k):
        vals = [MISSING if rnd.random() < 0.05 else round(rnd.uniform(-100, 100), 3) for _ in range(k)]
        cut = rnd.randint(0, k) if rnd.random() < 0.15 else k
        return vals[:cut] + [END] * (k - cut) if rnd.random() < 0.5 else vals[:max(cut, 1)]

    for i in range(n):
        pos += rnd.randint(0, 1000)
        na = rnd.choice([1, 2, 2, 2, 2, 3, 4])
        alleles = [b"ACGT"[rnd.randrange(4):][:1] * rnd.randint(1, 12) for _ in range(na)]
        if rnd.random() < 0.03:
            alleles[-1] = b""
        info = []
        for key in rnd.sample(["DP", "AF", "AC", "AN", "MQ", "DB", "SB", "ANN_S", "TAGS", "VALS", "FV"], rnd.randint(0, 8)):
            if key in ("DP", "AN"):
                tv = W.tv_ints(ints(1))
            elif key in ("AC", "VALS"):
                tv = W.tv_ints(ints(rnd.randint(1, 20)))
            elif key == "SB":
                tv = W.tv_ints(ints(4, 0, 300))
            elif key in ("AF", "FV"):
                tv = W.tv_floats(floats(rnd.randint(1, 5)))
            elif key == "MQ":
                tv = W.tv_floats(floats(1))
            elif key == "DB":
                tv = b"\x00"
            else:
                tv = W.tv_str(rnd.choice([b"", b".", b"x", b"a,b", b"," * rnd.randint(1, 3), bytes(rnd.choices(b"abc,.", k=rnd.randint(1, 40)))]))
            info.append((K[key], tv))
        fmt = []
        if ns:
            for key in rnd.sample(["GT", "GQ", "DP", "AD", "PL", "GL", "FT", "HQ", "SC"], rnd.randint(0, 9)):
                if key == "GT":
                    pl = rnd.choice([1, 2, 2, 2, 3])
                    per = []
                    for _ in range(ns):
                        g = [0 if rnd.random() < 0.05 else ((rnd.randint(0, na - 1) + 1) << 1) | rnd.randint(0, 1) for _ in range(rnd.randint(1, pl))]
                        per.append(g)
                    fmt.append(W.fmt_ints(K[key], per, n=pl))
                elif key in ("GQ", "DP"):
                    fmt.append(W.fmt_ints(K[key], [ints(1) for _ in range(ns)]))
                elif key in ("AD", "PL", "HQ"):
                    k = {"AD": na, "PL": na * (na + 1) // 2, "HQ": 2}[key]
                    fmt.append(W.fmt_ints(K[key], [ints(k) for _ in range(ns)], n=k))
                elif key == "GL":
                    k = na * (na + 1) // 2
                    fmt.append(W.fmt_floats(K[key], [floats(k) for _ in range(ns)], n=k))
                elif key == "SC":
                    fmt.append(W.fmt_floats(K[key], [floats(1) for _ in range(ns)]))
                else:
                    fmt.append(W.fmt_strs(K[key], [rnd.choice([b"PASS", b".", b"", b"q10;s50"]) for _ in range(ns)]))
        flt = rnd.choice([None, [0], [1], [1, 2], [2]])
        qual = None if rnd.random() < 0.05 else round(rnd.uniform(0, 5000), 1)
        ident = rnd.choice([b"", b".", b"rs%d" % rnd.randint(1, 10**9)])
        out.append(W.record(rnd.randrange(n_ctg), pos, len(alleles[0]), qual, ident, alleles, flt, info, fmt, ns))
    return out


def all_cases():
    """-> list of (name, file_bytes, tidy)"""
    ns = len(SAMPLES)
    hdr = std_header()
    good = basic_records()
    cases = []
    cases.append(("basic", W.bcf_bytes(hdr, good), False))
    cases.append(("basic_tidy", W.bcf_bytes(hdr, good), True))
    cases.append(("mismatch", W.bcf_bytes(hdr, mismatch_records()), False))
    cases.append(("qual_bits", W.bcf_bytes(std_header(samples=()), qual_records()), False))
    cases.append(("gt", W.bcf_bytes(hdr, gt_records()), False))
    cases.append(("gt_v44", W.bcf_bytes(std_header(fileformat="VCFv4.4"), gt_records()), True))
    cases.append(("empty_no_records", W.bcf_bytes(hdr, []), False))
    cases.append(("no_samples", W.bcf_bytes(std_header(samples=()), [W.record(0, 7, 1, 3.0, b"x", (b"A", b"T"), None, [(K["DP"], W.tv_ints([3]))])]), False))
    cases.append(("no_samples_tidy", W.bcf_bytes(std_header(samples=()), [W.record(0, 7, 1, 3.0, b"x", (b"A", b"T"))]), True))
    # samples but no FORMAT definitions: one default GT column, always NULL (bcf_reader.c:683-692)
    h2 = W.header(['##INFO=<ID=DP,Number=1,Type=Integer,Description="d">'], samples=("A", "B"), contigs=("1",))
    cases.append(("no_format_defs", W.bcf_bytes(h2, [W.record(0, 1, 1, 1.0, b"", (b"A",), None, [(1, W.tv_ints([4]))], [], 2)]), False))
    # IDX= dictionary: out of order, with holes; columns in ascending id order
    h3 = W.header(['##INFO=<ID=ZZ,Number=1,Type=Integer,Description="d",IDX=7>', '##INFO=<ID=AA1,Number=.,Type=String,Description="d",IDX=3>',
                   '##FILTER=<ID=PASS,Description="p",IDX=0>', '##FILTER=<ID=lq,Description="d",IDX=5>', '##FORMAT=<ID=GT,Number=1,Type=String,Description="d",IDX=9>',
                   '##FORMAT=<ID=ZZ,Number=2,Type=Float,Description="d",IDX=7>', '##INFO=<ID=NEW,Number=0,Type=Flag,Description="d">'],
                  samples=("A",), contigs=(("##contig=<ID=c5,length=10,IDX=5>",), ("##contig=<ID=c1,IDX=1>",)))
    cases.append(("idx_header", W.bcf_bytes(h3, [
        W.record(5, 1, 1, 1.0, b"a", (b"A",), [5], [(7, W.tv_ints([1])), (3, W.tv_str(b"p,q")), (10, b"\x00")], [W.fmt_ints(9, [W.gt(0, 0)]), W.fmt_floats(7, [[1.0, 2.0]])], 1),
        W.record(1, 2, 1, 1.0, b"b", (b"A",), None, [], [], 1),
        W.record(0, 3, 1, 1.0, b"hole-contig", (b"A",), None, [], [], 1)]), False))
    # spec corrections (vcf_types.h:119-205): AC Number=1 -> list, DP Number=. -> scalar, FORMAT GQ Number=. -> scalar, AD Number=. stays list, GT Number=. -> scalar
    h4 = W.header(['##INFO=<ID=AC,Number=1,Type=Integer,Description="d">', '##INFO=<ID=DP,Number=.,Type=Integer,Description="d">',
                   '##INFO=<ID=AF,Number=.,Type=Float,Description="d">', '##INFO=<ID=SB,Number=.,Type=Integer,Description="d">',
                   '##INFO=<ID=H2,Number=1,Type=Flag,Description="d">', '##INFO=<ID=CH,Number=1,Type=Character,Description="d">', '##INFO=<ID=NOTYPE,Description="d">',
                   '##FORMAT=<ID=GQ,Number=.,Type=Integer,Description="d">', '##FORMAT=<ID=AD,Number=.,Type=Integer,Description="d">',
                   '##FORMAT=<ID=GT,Number=.,Type=String,Description="d">', '##FORMAT=<ID=PL,Number=3,Type=Integer,Description="d">',
                   '##FORMAT=<ID=XS,Number=.,Type=String,Description="d">', '##FORMAT=<ID=FL,Number=0,Type=Flag,Description="d">'],
                  samples=("A", "B"), contigs=("1",))
    cases.append(("spec_corrections", W.bcf_bytes(h4, [
        W.record(0, 1, 1, 1.0, b"", (b"A", b"C", b"G"), None,
                 [(1, W.tv_ints([3, 4])), (2, W.tv_ints([9, 8])), (3, W.tv_floats([0.1, 0.2])), (4, W.tv_ints([1, 2, 3, 4])), (5, b"\x00"), (6, W.tv_str(b"c")), (7, W.tv_str(b"u,v"))],
                 [W.fmt_ints(8, [[5, 6], [7]]), W.fmt_ints(9, [[1, 2, 3], [4]]), W.fmt_ints(10, [W.gt(0, 1), W.gt(1, 2, phased=True)]), W.fmt_ints(11, [[1, 2, 3], [4, 5, 6]]),
                  W.fmt_strs(12, [b"a,b", b"c"]), W.fmt_ints(13, [[1], [1]])], 2)]), False))
    # header GT declared Integer: FORMAT_GT column INTEGER and always NULL (getter type check vcf.c:6183-6187)
    h5 = W.header(['##FORMAT=<ID=GT,Number=1,Type=Integer,Description="d">'], samples=("A",), contigs=("1",))
    cases.append(("gt_integer_header", W.bcf_bytes(h5, [W.record(0, 1, 1, 1.0, b"", (b"A", b"C"), None, [], [W.fmt_ints(1, [W.gt(0, 1)])], 1)]), False))

    # ---- streams that end on a bad record: rows before it are kept ------------------------------------------------
    def bad(name, rec):
        cases.append((name, W.bcf_bytes(hdr, good[:2] + [rec] + good[2:]), False))
    bad("bad_rid", W.record(3, 1, 1, 1.0, b"x", (b"A",), n_sample=ns))
    bad("bad_rid_negative", W.record(-1, 1, 1, 1.0, b"x", (b"A",), n_sample=ns))
    bad("bad_no_allele", W.record(0, 1, 1, 1.0, b"x", (), n_sample=ns))
    bad("bad_filter_key", W.record(0, 1, 1, 1.0, b"x", (b"A",), [99], n_sample=ns))
    bad("bad_filter_type", W.record(0, 1, 1, 1.0, b"x", (b"A",), filter_raw=W.tv_floats([1.0]), n_sample=ns))
    bad("bad_info_key", W.record(0, 1, 1, 1.0, b"x", (b"A",), None, [(77, W.tv_ints([1]))], n_sample=ns))
    bad("bad_info_type", W.record(0, 1, 1, 1.0, b"x", (b"A",), None, [(K["DP"], bytes([0x14]) + b"\0" * 8)], n_sample=ns))
    bad("bad_info_null_with_len", W.record(0, 1, 1, 1.0, b"x", (b"A",), None, [(K["DP"], bytes([0x10]))], n_sample=ns))
    bad("bad_fmt_key", W.record(0, 1, 1, 1.0, b"x", (b"A",), None, [], [W.fmt_ints(99, [[1]] * ns)], ns))
    bad("bad_fmt_short", W.record(0, 1, 1, 1.0, b"x", (b"A",), None, [], [W.fmt_ints(K["GQ"], [[1]] * (ns - 1))], ns))
    bad("bad_gt_short", W.record(0, 1, 1, 1.0, b"x", (b"A",), None, [], [W.fmt_ints(K["GT"], [[2, 4]] * (ns - 1))], ns))
    bad("bad_n_info_overrun", W.record(0, 1, 1, 1.0, b"x", (b"A",), None, [(K["DP"], W.tv_ints([1]))], n_info=3, n_sample=ns))
    bad("bad_l_shared", struct.pack("<II", 20, 0) + b"\0" * 24)
    idrec = bytearray(W.record(0, 1, 1, 1.0, b"x", (b"A",), n_sample=ns))
    idrec[32] = 0x11                                       # ID stored as int8
    bad("bad_id_type", bytes(idrec))
    alrec = bytearray(W.record(0, 1, 1, 1.0, b"", (b"A",), n_sample=ns))
    alrec[33] = 0x11                                       # allele stored as int8
    bad("bad_allele_type", bytes(alrec))
    raw = W.bcf_raw(hdr, good)
    cases.append(("truncated_mid_record", W.bgzf_file(raw[:-7]), False))
    cases.append(("truncated_core", W.bgzf_file(raw + b"\x40\0\0\0\0\0\0\0\1\2\3"), False))
    # n_fmt > 0 but no samples / no indiv bytes: n_fmt is silently zeroed (vcf.c:1906)
    cases.append(("nfmt_without_indiv", W.bcf_bytes(hdr, [W.record(0, 1, 1, 1.0, b"x", (b"A",), n_fmt=3, n_sample=ns), W.record(0, 2, 1, 1.0, b"y", (b"A",), None, [], [W.fmt_ints(K["GQ"], [[1]] * ns)], 0)]), False))
    # fewer samples in the record than in the header: cells beyond the record's n_sample are NULL here (parity domain note)
    cases.append(("nsample_less_than_header", W.bcf_bytes(hdr, [W.record(0, 1, 1, 1.0, b"x", (b"A",), None, [], [W.fmt_ints(K["GQ"], [[1], [2]])], 2)]), False))

    # ---- framing across BGZF blocks, and volume ------------------------------------------------------------------------
    fz = fuzz_records(7, 3000, ns)
    cases.append(("fuzz_small_blocks", W.bcf_bytes(hdr, fz, payload=777), False))
    cases.append(("fuzz_tidy", W.bcf_bytes(hdr, fz[:800], payload=4000), True))
    cases.append(("fuzz_sites_only", W.bcf_bytes(std_header(samples=()), fuzz_records(8, 3000, 0)), False))
    big_al = [W.record(0, 1, 1, 1.0, b"big", (b"A" * 70000, b"C" * 300), None, [(K["VALS"], W.tv_ints(list(range(40000))))], [W.fmt_ints(K["PL"], [list(range(300))] * ns)], ns)]
    cases.append(("long_record", W.bcf_bytes(hdr, good[:1] + big_al + good[1:]), False))
    return cases


def header_error_cases():
    hdr = std_header()
    raw = W.bcf_raw(hdr, basic_records())
    out = [("not_bgzf", b"hello world, definitely not a BGZF file" * 4), ("bad_magic", W.bgzf_file(b"BCF\x02\x01" + raw[5:])),
           ("bam_magic", W.bgzf_file(b"BAM\x01" + raw[4:])), ("truncated_header", W.bgzf_file(raw[:40])),
           ("no_chrom_line", W.bgzf_file(W.bcf_raw("##fileformat=VCFv4.2\n##contig=<ID=1>\n", []))),
           ("dup_sample", W.bgzf_file(W.bcf_raw(W.header([], samples=("A", "A"), contigs=("1",)), []))),
           ("idx_conflict", W.bgzf_file(W.bcf_raw(W.header(['##INFO=<ID=A,Number=1,Type=Integer,Description="d",IDX=1>', '##INFO=<ID=B,Number=1,Type=Integer,Description="d",IDX=1>'], contigs=("1",)), [])))]
    return out
