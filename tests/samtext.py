"""The reference's SAM TEXT fixtures as BAM: read_bam here takes BAM only, so tests that want to stand on test/data/rg.sam.gz and
test/data/aux_tags.sam.gz (duckhts.test:164-185) re-encode the fixture's own lines with tests/bamwriter.py -- what samtools view -b
would do (sam_parse1, htslib sam.c:2657-2838: '=' for RNEXT, QUAL '*' -> 0xff, integer tags in the smallest type that holds them)."""
import gzip
import os

import bamwriter as W

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _tag(t):
    name, ty, val = t.split(":", 2)
    if ty == "i":
        v = int(val)
        ty = ("c" if v >= -128 else "s" if v >= -32768 else "i") if v < 0 else ("C" if v <= 255 else "S" if v <= 65535 else "I")
        return (name, ty, v)
    if ty == "f":
        return (name, "f", float(val))
    if ty == "B":
        sub, *vals = val.split(",")
        return (name, "B:" + sub, [float(x) if sub == "f" else int(x) for x in vals])
    return (name, ty, val)


def sam_fixture_as_bam(name, **kw):
    """-> BAM file bytes made from the lines of tests/golden/<name> (a gzip / BGZF compressed SAM text file)"""
    return sam_text_as_bam(gzip.open(os.path.join(GOLD, name), "rb").read().decode(), **kw)


def sam_text_as_bam(text, **kw):
    hdr = "".join(l + "\n" for l in text.split("\n") if l.startswith("@"))
    refs = []
    for l in hdr.split("\n"):
        if l.startswith("@SQ"):
            f = dict(x.split(":", 1) for x in l.split("\t")[1:])
            refs.append((f["SN"], int(f["LN"])))
    names = [n for n, _ in refs]
    recs = []
    for l in text.split("\n"):
        if not l or l.startswith("@"):
            continue
        f = l.split("\t")
        tid = -1 if f[2] == "*" else names.index(f[2])
        mtid = tid if f[6] == "=" else -1 if f[6] == "*" else names.index(f[6])
        recs.append(W.record(qname=f[0], flag=int(f[1]), tid=tid, pos=int(f[3]) - 1, mapq=int(f[4]), cigar=f[5], mtid=mtid, mpos=int(f[7]) - 1, tlen=int(f[8]),
                             seq=f[9], qual=None if f[10] == "*" else f[10], tags=[_tag(t) for t in f[11:]]))
    return W.bam_bytes(refs, recs, text=hdr, **kw)
