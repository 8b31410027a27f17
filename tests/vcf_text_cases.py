"""Sites-only VCF TEXT inputs (plain and BGZF) shared by the oracle tests (CPU) and the GPU parity tests."""
import random

import bamwriter as W

HDR = [
    "##fileformat=VCFv4.2",
    '##FILTER=<ID=q10,Description="Quality below 10">',
    '##FILTER=<ID=s50,Description="x">',
    "##contig=<ID=chr1,length=1000000>",
    "##contig=<ID=chr2>",
    '##INFO=<ID=DP,Number=1,Type=Integer,Description="d">',
    '##INFO=<ID=AF,Number=A,Type=Float,Description="d">',
    '##INFO=<ID=AC,Number=A,Type=Integer,Description="d">',
    '##INFO=<ID=MQ,Number=1,Type=Float,Description="d">',
    '##INFO=<ID=DB,Number=0,Type=Flag,Description="d">',
    '##INFO=<ID=SB,Number=4,Type=Integer,Description="d">',
    '##INFO=<ID=ANN_S,Number=1,Type=String,Description="d">',
    '##INFO=<ID=TAGS,Number=.,Type=String,Description="d">',
    '##INFO=<ID=FV,Number=.,Type=Float,Description="d">',
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="only a FORMAT definition: not an INFO key">',
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO",
]


def bamwriter_bgzf(raw, **kw):
    return W.bgzf_file(raw, **kw)


def text(lines, hdr=HDR, eol="\n", last_eol=True):
    s = eol.join(list(hdr) + list(lines))
    return (s + (eol if last_eol else "")).encode()


def L(chrom="chr1", pos="100", vid=".", ref="A", alt="T", qual="30", flt="PASS", info="."):
    return "\t".join([chrom, str(pos), vid, ref, alt, qual, flt, info])


NUMBERS = [
    L(pos=1, info="DP=35;AF=0.5;AC=2;MQ=59.5;DB;SB=1,2,3,4;ANN_S=missense;TAGS=a,b,,c,;FV=1.5,2.25"),
    L(pos=2, qual=".", info="DP=+5;AC=-,+,12abc,,7;MQ=.5;FV=5.,-0.0,00012.500,1e-3,1E2,nan,inf,-inf,0x10,abc,,."),
    L(pos=3, qual="12abc", info="DP=99999999999;AC=2147483647,-2147483640,-2147483641,2147483648,99999999999999999999,-99999999999999999999"),
    L(pos=4, qual="1e2", info="DP=.;AF=.,0.25;MQ=123456789012345.5;FV=0.1234567890123,0.12345678901234,0.123456789012345,1234567890123456789"),
    L(pos=5, qual=" 7.25", info="MQ= 3.5;FV=+1.5,-2.5, 4,1 ,0.000000000000001,100000000000000"),
    L(pos=6, qual="-", info="DP=;AF=;ANN_S=;TAGS=;DB=yes;MQ"),
    L(pos=7, qual="inf", info="FV=3.4028235e38,3.5e38,1e-46,4.9e-324"),
    L(pos=8, qual="0.1", info="FV=0.1,0.2,0.3,16777217,0.30000000000000004"),
    L(pos=9, qual="3.98e-06", info="MQ=1.5e3;FV=3.98e-06,1.23456e-05,1e22,1e23,123456789012345e8,1e-22,1e-23,5e-324,1.7976931348623157e308,1e,1e+,1.e2,.5e1,0.e0,00.00e5,1E-0,9.99999999999999e22"),
    L(pos=10, qual=".5", info="FV=.5,-.25,+.125,.,-.,.e5,0.000000000000000000001,1000000000000000000000,0.1e-21,12345678901234e9"),
]
FIELDS = [
    L(pos=10, vid="rs1;rs2", ref="ACGT", alt="A,<DEL>,ACGTT"),
    L(pos=11, vid="", ref="", alt=""),
    L(pos=12, alt="."),
    L(pos=13, alt="A,"),
    L(pos=14, alt=",A"),
    L(pos=15, flt="."),
    L(pos=16, flt="q10;s50"),
    L(pos=17, flt="q10;"),
    L(pos=18, flt="q10;;s50"[0:3]),
    L(pos=19, flt="DP"),                        # a name defined only as INFO is still in the dictionary
    L(pos=20, info="DP=1;;AF=0.5"),
    L(pos=21, info=";DP=2"),
    L(pos=22, info="=5;DP=3"),
    L(pos=23, info="ANN_S=a=b=c;DP=4"),
    L(pos=24, info="DP=5;"),
    L(pos=25, info="DP=6;DP=7"),
    L(pos=0, info="DP=8"),
    L(pos="+30", info="DP=9"),
    L(chrom="chr2", pos=2147483646),
]
UNDEFINED = [
    L(chrom="chrUn_1", pos=1, flt="lowq", info="NEWKEY=5;DP=1"),
    L(chrom="chr2", pos=2, flt="lowq;other", info="NEWFLAG;GT=x;NEWKEY=a,b"),
    L(chrom="chrUn_2", pos=3, flt="PASS", info="q10=1"),            # a FILTER name used as an INFO key: gets an INFO definition
    L(chrom="chrUn_1", pos=4, flt="other", info="NEWKEY=.;AF=0.5"),
]


SHDR = HDR[:-1] + [
    '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="d">',
    '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="d">',
    '##FORMAT=<ID=GL,Number=G,Type=Float,Description="d">',
    '##FORMAT=<ID=FT,Number=1,Type=String,Description="d">',
    '##FORMAT=<ID=FLG,Number=0,Type=Flag,Description="a FORMAT Flag is not supported by the parser">',
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\tS3",
]


def S(fmt, *samples, pos=100, info="."):
    return L(pos=pos, info=info) + "\t" + "\t".join((fmt,) + samples)


SAMPLES = [
    S("GT:GQ:AD:GL:FT", "0/1:99:10,20:-0.1,-2.5,-30:PASS", "1|1:45:0,300:-1,-2,-3:lowq;x", "./.:.:.,.:.:.", pos=1),
    S("GT:GQ", "0|1|2:5", ".:7", "1:.", pos=2),                                # ploidy 3, haploid
    S("GT", "0", "1/2", ".|.", pos=3),
    S("GQ:GT", "5:0/1", "6", "7:1/1", pos=4),                                  # GT not first; a sample that leaves the trailing field out
    S("GT:AD:GL", "0/1:1,2,3:.5,1e-3,5.", "1/1:.:nan,inf", "0/0:7:1E2", pos=5),  # strtod forms in FORMAT floats
    S("FT:GQ", "abc:1", ":2", "a:", pos=6),                                    # empty string, empty integer
    S("GL", "", "1", ".,.", pos=7),                                            # an empty Float value stores 0.0
    S("GT:GQ", "10/11:+5", "1/1:-", "0/0:2147483647", pos=8),
    S(".", "whatever", "x", "y", pos=9),                                       # FORMAT ".": no fields
    L(pos=10),                                                                 # eight columns only: no FORMAT at all
    S("GT:NEWF:GQ", "0/1:xyz:3", "0/0:q:4", "1/1:r,s:5", pos=11),              # an undefined FORMAT key becomes a String
    S("GT:GT", "0/1:1/1", "0/0:0/1", "1/1:0/0", pos=12),                       # duplicate tag: the second is dropped
    S("GT:GQ", "0/1:5", "0/0:6", "1/1:7", "0/0:8", pos=13),                    # more columns than samples: ignored
    S("GT:AD", "0/1:1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17", "1/1:1", "0/0:.", pos=14),
]
SAMPLE_ERRORS = {"fewer_columns_than_samples": S("GT", "0/1", "0/0"), "more_fields_than_format": S("GT", "0/1:5", "0/0", "1/1"), "gt_not_a_number": S("GT", "0/x", "0/0", "1/1"),
                 "int_with_garbage": S("GQ", "12abc", "1", "2"), "float_with_garbage": S("GL", "1.5x", "1", "2"), "float_exponent_with_garbage": S("GL", "1e5x", "1", "2"),
                 "format_flag": S("FLG", "1", "1", "1"), "format_dot_key": S("GT:.", "0/1:1", "0/0:1", "1/1:1"), "format_without_samples": L() + "\tGT", "gt_empty": S("GT", "", "0/0", "1/1")}


def all_cases():
    """-> list of (name, file bytes)"""
    out = [("numbers_plain", text(NUMBERS)), ("numbers_bgzf", W.bgzf_file(text(NUMBERS))),
           ("fields", text(FIELDS)), ("fields_crlf_no_final_eol", text(FIELDS, eol="\r\n", last_eol=False)),
           ("undefined_names", text(UNDEFINED)), ("undefined_names_bgzf_small_blocks", W.bgzf_file(text(UNDEFINED * 40), payload=700)),
           # names of 5, 6, 7, 13, 14 and 15 bytes that differ only in their FIRST bytes (the device compares names 8 bytes at a time and
           # then the rest: soak seeds 3037 / 3086 found the rest compared through a 32-bit accumulator, i.e. by its last four bytes)
           ("undefined_names_same_tails", text([L(chrom=c, pos=i + 1, flt=f, info=k + "=1") for i, (c, f, k) in enumerate(
               [("chrUn1", "lowq7", "NEWKEY1"), ("c8rUn1", "Xowq7", "NxWKEY1"), ("2hrUn1", "lXwq7", "XEWKEY1"), ("chrUn1", "lowq7", "NEWKEY1"), ("xbrUn", "loXq7x", "NEWKEY1"),
                ("abrUn", "lowq7x", "NEXKEY1"), ("chr_unplaced_1", "filter_name_15", "INFO_KEY_13_"), ("chr_unplXced_1", "fXlter_name_15", "INFO_XEY_13_"),
                ("Xhr_unplaced_1", "filter_nXme_15", "XNFO_KEY_13_"), ("c8rUn1", "Xowq7", "NxWKEY1"), ("chr_unplaced_1", "filter_name_15", "INFO_KEY_13_")])]))]
    # positions beyond 32 bits: hts_pos_t is 64 bits wide and vcf_parse takes up to 62 (vcf.c:4052-4063): POS 0 (the telomere: -1), the edges of the
    # 32-bit words, END= / SVLEN= behind them, a sample column
    out.append(("positions_beyond_32_bits", text([L(pos=p, info=i) for p, i in ((0, "."), (2147483647, "DP=1"), (2147483648, "END=2147483999"), (4294967295, "."), (4294967296, "DP=2"), (4294967297, "END=4294968000"),
                                                                                (1 << 40, "DP=3"), ((1 << 62) - 1, "."), (1 << 62, "DP=4"), (5, "."))])))
    for name, bad in (("bad_pos", L(pos="1x")), ("bad_pos_overflow", L(pos="9223372036854775807")), ("pos_beyond_62_bits", L(pos=(1 << 62) + 1)), ("too_few_columns", "chr1\t5\t.\tA\tT\t.\tPASS"),
                      ("empty_line", ""), ("undefined_contig_with_bad_name", L(chrom="a,b>")), ("empty_filter_name", L(flt="q10;;s50"))):
        out.append((name, text(FIELDS[:3] + [bad] + FIELDS[3:6])))
    out.append(("nul_in_line", text([L(pos=1, info="DP=1"), L(pos=2, info="DP=2\0;AF=0.5"), L(pos=3, vid="a\0b", info="DP=3")])))
    out.append(("ninth_column_ignored", text([L(pos=1, info="DP=1") + "\tGT\t0/1", L(pos=2)])))
    out.append(("header_with_empty_lines_and_no_filters", text([L(pos=1, flt="PASS")], hdr=["##fileformat=VCFv4.1", "", "##contig=<ID=chr1>", "", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"])))
    out.append(("header_only", text([])))
    out.append(("samples", text(SAMPLES, hdr=SHDR)))
    out.append(("samples_v44_bgzf", bamwriter_bgzf(text([S("GT:GQ", "/0/1:1", "|1|0:2", "0:3", pos=1), S("GT", "|0", "/1", ".", pos=2)], hdr=["##fileformat=VCFv4.4"] + SHDR[1:]))))
    for name, bad in SAMPLE_ERRORS.items():
        out.append(("samples_" + name, text(SAMPLES[:2] + [bad] + SAMPLES[2:4], hdr=SHDR)))
    rs = random.Random(11)
    big = []
    for i in range(3000):
        smp = []
        for _ in range(3):
            gt = rs.choice(["0/0", "0/1", "1|1", "./.", "1/2", "0"])
            smp.append(":".join([gt, rs.choice([".", str(rs.randrange(100))]), ",".join(rs.choice([".", str(rs.randrange(300))]) for _ in range(rs.randrange(1, 4))),
                                 ",".join(rs.choice([".", "%.3f" % -rs.random(), "%g" % -(rs.random() * 50)]) for _ in range(rs.randrange(1, 4))), rs.choice(["PASS", ".", "lowq"])][:rs.randrange(1, 6)]))
        big.append(S("GT:GQ:AD:GL:FT", *smp, pos=1 + i, info=rs.choice([".", "DP=%d" % i, "AF=0.5;DB"])))
    out.append(("samples_many_bgzf", bamwriter_bgzf(text(big, hdr=SHDR), payload=4000)))
    rnd = random.Random(7)
    many = []
    for i in range(6000):
        info = ";".join(rnd.sample(["DP=%d" % rnd.randrange(-5, 100000), "AF=%s" % ",".join("%.*f" % (rnd.randrange(0, 9), rnd.random()) for _ in range(rnd.randrange(1, 4))),
                                    "AC=%s" % ",".join(str(rnd.randrange(0, 70000)) for _ in range(rnd.randrange(1, 4))), "MQ=%g" % (rnd.random() * 10 ** rnd.randrange(-8, 9)), "DB",
                                    "SB=1,2,3,4", "ANN_S=" + "x" * rnd.randrange(0, 400), "TAGS=" + ",".join("t%d" % rnd.randrange(100) for _ in range(rnd.randrange(0, 5))),
                                    "FV=" + ",".join(rnd.choice([".", "1e-%d" % rnd.randrange(0, 50), "%d" % rnd.randrange(0, 10 ** rnd.randrange(1, 19)), "%.17g" % rnd.random()]) for _ in range(rnd.randrange(1, 5)))],
                                   rnd.randrange(0, 7))) or "."
        many.append(L(chrom=rnd.choice(["chr1", "chr2"]), pos=1 + i, vid=rnd.choice([".", "rs%d" % i]), ref=rnd.choice(["A", "ACGT" * rnd.randrange(1, 30)]),
                      alt=rnd.choice([".", "T", "T,G", "<DEL>"]), qual=rnd.choice([".", "%.2f" % (rnd.random() * 1000), "%d" % rnd.randrange(1000), "%g" % (rnd.random() * 1e-5)]),
                      flt=rnd.choice(["PASS", ".", "q10", "q10;s50"]), info=info))
    out.append(("many_plain", text(many)))
    out.append(("many_bgzf", W.bgzf_file(text(many), payload=3000)))
    return out
