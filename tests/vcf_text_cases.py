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


def all_cases():
    """-> list of (name, file bytes)"""
    out = [("numbers_plain", text(NUMBERS)), ("numbers_bgzf", W.bgzf_file(text(NUMBERS))),
           ("fields", text(FIELDS)), ("fields_crlf_no_final_eol", text(FIELDS, eol="\r\n", last_eol=False)),
           ("undefined_names", text(UNDEFINED)), ("undefined_names_bgzf_small_blocks", W.bgzf_file(text(UNDEFINED * 40), payload=700))]
    for name, bad in (("bad_pos", L(pos="1x")), ("bad_pos_overflow", L(pos="9223372036854775807")), ("pos_too_large_for_bcf", L(pos=2147483648)), ("too_few_columns", "chr1\t5\t.\tA\tT\t.\tPASS"),
                      ("empty_line", ""), ("undefined_contig_with_bad_name", L(chrom="a,b>")), ("empty_filter_name", L(flt="q10;;s50"))):
        out.append((name, text(FIELDS[:3] + [bad] + FIELDS[3:6])))
    out.append(("nul_in_line", text([L(pos=1, info="DP=1"), L(pos=2, info="DP=2\0;AF=0.5"), L(pos=3, vid="a\0b", info="DP=3")])))
    out.append(("ninth_column_ignored", text([L(pos=1, info="DP=1") + "\tGT\t0/1", L(pos=2)])))
    out.append(("header_with_empty_lines_and_no_filters", text([L(pos=1, flt="PASS")], hdr=["##fileformat=VCFv4.1", "", "##contig=<ID=chr1>", "", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"])))
    out.append(("header_only", text([])))
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
