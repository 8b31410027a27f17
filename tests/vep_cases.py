"""VEP / CSQ inputs shared by the oracle tests (CPU) and the GPU parity tests.

tests/golden/test_vep.vcf.gz is the reference's own fixture test/data/test_vep.vcf (data, gzip-compressed as is): 802 sites-only
records with INFO/AF and an 80-field INFO/CSQ.  read_bcf here takes BCF, so the text is re-encoded record by record with
tests/bcfwriter.py (the reference reads the text form through htslib's VCF parser, which builds the same bcf1_t).
"""
import gzip
import os
import struct

import numpy as np

import bcfwriter as W

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def fixture_text():
    with gzip.open(os.path.join(GOLD, "test_vep.vcf.gz"), "rb") as f:
        return f.read().decode()


def fixture_bcf():
    """-> (bcf file bytes, header field names, list of (pos, csq string or None)) for the reference fixture"""
    lines = fixture_text().splitlines()
    hdr = [l for l in lines if l.startswith("#")]
    ids = {"PASS": 0}                                     # dictionary ids: PASS, then order of first appearance
    for l in hdr:
        if l.startswith("##INFO=<ID=") or l.startswith("##FILTER=<ID=") or l.startswith("##FORMAT=<ID="):
            ids.setdefault(l.split("ID=")[1].split(",")[0], len(ids))
    desc = [l for l in hdr if l.startswith("##INFO=<ID=CSQ")][0]
    fields = desc.split("Format: ")[1].split('"')[0].split("|")
    recs, rows = [], []
    for l in lines:
        if l.startswith("#") or not l:
            continue
        chrom, pos, vid, ref, alt, qual, flt, info = l.split("\t")[:8]
        assert chrom == "1" and flt == "PASS"
        inf, csq = [], None
        for kv in info.split(";"):
            k, _, v = kv.partition("=")
            if k == "AF":
                inf.append((ids["AF"], W.tv_floats([float(np.float32(float(x))) for x in v.split(",")])))
            elif k == "CSQ":
                csq = v
                inf.append((ids["CSQ"], W.tv_str(v.encode())))
            else:
                raise AssertionError(k)
        recs.append(W.record(0, int(pos) - 1, len(ref), float(np.float32(float(qual))) if qual != "." else None, b"" if vid == "." else vid.encode(),
                             (ref.encode(),) + tuple(a.encode() for a in alt.split(",")), [0], inf))
        rows.append((int(pos), csq))
    return W.bcf_bytes("\n".join(hdr) + "\n", recs), fields, rows


def py_split(csq, n_fields):
    """independent reading of one CSQ value: list over transcripts of n_fields tokens (None = missing); None when there is no transcript"""
    if csq is None:
        return None
    trs = [t for t in csq.split(",") if t != ""]
    if not trs:
        return None
    out = []
    for t in trs:
        toks = [x.strip(" \t\n\v\f\r") for x in t.split("|")][:n_fields]
        toks += [""] * (n_fields - len(toks))
        out.append([None if x in ("", ".") else x for x in toks])
    return out


FMT = "Allele|Consequence|SYMBOL|DISTANCE|STRAND|gnomAD_AF|MAX_AF|MOTIF_POS|FLAGS|SpliceAI_pred_DS_AG|NOTE"
FIELDS = FMT.split("|")


def _hdr(tag="CSQ", typ="String", fmt=FMT, samples=(), extra=()):
    lines = ['##INFO=<ID=DP,Number=1,Type=Integer,Description="d">',
             f'##INFO=<ID={tag},Number=.,Type={typ},Description="Consequence annotations from Ensembl VEP. Format: {fmt}">',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="d">'] + list(extra)
    return W.header(lines, samples=samples, contigs=("chr1", "chr2"))


# dictionary ids for _hdr(): PASS 0, DP 1, <tag> 2, GT 3
VALUES = [
    b"T|missense_variant&splice|GENE1|12|-1|0.25|1e-3|7|cds_start_NF|0.5|note one",
    b"T|a|b|3|1|.5|5.|+9||1E2|x,G|c|d|-4|+1|-0.0|inf|0|f|nan|y",                                        # two transcripts
    b",,T|only_two",                                                                                   # empty pieces skipped; missing tail fields
    b",,,",                                                                                            # no transcript at all -> NULL row
    b".",                                                                                              # one transcript "." -> every field missing
    b" T | . |  | 12x | 1 2 |abc|0x10|99999999999999999999|\t|1e400|  spaced  out  ",                   # trimming, bad numbers, overflow
    b"T|a|b|-99999999999999999999|2147483648|1e-50|-|.|.|.|.|extra|fields|dropped,|||||||||||",          # more fields than declared; all-empty transcript
    b"A|b|c|007|-0|  3.25|4,",                                                                         # trailing comma
    b"T|x\0|hidden",                                                                                   # C string: cut at the NUL
    b"|",
    b"T|a|b|-|+|-.|e5|1e|z|0x1p-2|q",
]


def edge_cases():
    """-> list of (name, file bytes, tidy)"""
    out = []
    recs = [W.record(0, 10 + i, 1, 30.0, b"", (b"A", b"T"), None, [(1, W.tv_ints([i])), (2, W.tv_str(v))]) for i, v in enumerate(VALUES)]
    recs.append(W.record(0, 500, 1, 30.0, b"", (b"A", b"T"), None, [(1, W.tv_ints([5]))]))                      # tag absent
    recs.append(W.record(0, 501, 1, 30.0, b"", (b"A", b"T"), None, [(2, W.desc(0, 7))]))                       # zero-length value
    recs.append(W.record(0, 502, 1, 30.0, b"", (b"A", b"T"), None, [(2, W.tv_ints([65, 124, 66]))]))           # int8 vector under a String tag: read as bytes "A|B"
    out.append(("csq_values", W.bcf_bytes(_hdr(), recs), False))
    for tag in ("BCSQ", "ANN", "VEP", "vep"):
        out.append((f"tag_{tag}", W.bcf_bytes(_hdr(tag=tag), recs[:3]), False))
    out.append(("declared_integer", W.bcf_bytes(_hdr(typ="Integer"), recs[:2] + [recs[-1]]), False))           # bcf_get_info_string refuses: every row NULL
    out.append(("no_format_in_description", W.bcf_bytes(_hdr(fmt="").replace(" Format: ", " "), recs[:2]), False))   # no VEP columns at all
    out.append(("one_field", W.bcf_bytes(_hdr(fmt="Allele"), recs[:4]), False))
    out.append(("format_to_end_of_value", W.bcf_bytes(_hdr(fmt="A|B_AF|").replace('|">', '|>').replace('Description="Cons', 'Description=Cons'), recs[:2]), False))
    # two annotation tags: CSQ wins over ANN whatever the order of the header lines (vep_detect_tag)
    extra = ['##INFO=<ID=CSQ,Number=.,Type=String,Description="x Format: P|Q">']
    r2 = [W.record(0, 10, 1, 30.0, b"", (b"A", b"T"), None, [(2, W.tv_str(b"ann1|ann2")), (4, W.tv_str(b"csq1|csq2,c|d"))])]
    out.append(("csq_beats_ann", W.bcf_bytes(_hdr(tag="ANN", extra=extra), r2), False))
    # samples: wide and tidy (only a record's first sample row carries the annotation)
    smp = ("S1", "S2", "S3")
    rs = [W.record(0, 10 + i, 1, 30.0, b"", (b"A", b"T"), None, [(2, W.tv_str(v))], [W.fmt_ints(3, [W.gt(0, 1), W.gt(1, 1), W.gt(0, 0)])], 3) for i, v in enumerate(VALUES[:5])]
    out.append(("samples_wide", W.bcf_bytes(_hdr(samples=smp), rs), False))
    out.append(("samples_tidy", W.bcf_bytes(_hdr(samples=smp), rs), True))
    # many records over several BGZF blocks
    many = [W.record(0, 1000 + i, 1, 30.0, b"", (b"A", b"T"), None, [(1, W.tv_ints([i])), (2, W.tv_str(VALUES[i % len(VALUES)] * (1 + i % 3)))]) for i in range(3000)]
    out.append(("many", W.bcf_bytes(_hdr(), many), False))
    return out
