"""htslib's own SAM fixtures (third_party/htslib/test/*.sam, committed as data under tests/golden/htslib_sam/) through read_bam.

SURVEY 8(c) lists them as the vectors that pin the record semantics: auxf#values.sam (every aux type and integer width), xx#large_aux.sam /
xx#large_aux2.sam (hundreds of tags per record), c1#bounds / clip / noseq / pad1-3 / unknown (SEQ and QUAL present or '*', padded and
clipped CIGARs, mapped and unmapped forms), ce#unmap*.sam (unplaced reads, no @SQ).  The files are SAM text, which read_bam here does not
take: tests/samtext.py re-encodes their own lines as BAM the way sam_parse1 does (sam.c:2657-2838), and every expectation below is read
from the SAM TEXT ITSELF, not from the oracle: the 13 core columns as the reference's writers render them (bam_reader.c:785-877: RNEXT is a
name, never '='), READ_GROUP_ID / SAMPLE_ID from the RG tag and the @RG line, the typed standard-tag columns (bam_reader.c:54-70, 920-966)
and the AUXILIARY_TAGS map with bam_aux_to_string's text (bam_reader.c:140-183: %lld, %g, 'subtype,v,v,...')."""
import gzip
import os
import struct

import pytest

import orc
import samtext

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ["auxf#values.sam", "xx#large_aux.sam", "xx#large_aux2.sam", "c1#bounds.sam", "c1#clip.sam", "c1#noseq.sam", "c1#pad1.sam",
            "c1#pad2.sam", "c1#pad3.sam", "c1#unknown.sam", "ce#unmap.sam", "ce#unmap1.sam", "ce#unmap2.sam"]


def _text(name):
    return gzip.open(os.path.join(HERE, "golden", "htslib_sam", name + ".gz"), "rb").read().decode()


def _g(x):
    """printf("%g") of a value held as float32"""
    return ("%g" % struct.unpack("<f", struct.pack("<f", float(x)))[0]).encode()


def _aux_text(ty, val):
    if ty == "f":
        return _g(val)
    if ty == "B":
        sub, *vals = val.split(",")
        return (sub + "".join("," + (_g(v).decode() if sub == "f" else str(int(v))) for v in vals)).encode()
    if ty == "i":
        return str(int(val)).encode()
    return val.encode()


def expectations(text, std):
    """the table the reference would show for this SAM text: core columns, RG / SM, typed standard tags, the auxiliary-tag map"""
    std_by = {n: (t, s) for n, t, s in std}
    sm_of = {}
    for l in text.split("\n"):
        if l.startswith("@RG"):
            f = dict(x.split(":", 1) for x in l.split("\t")[1:])
            if "ID" in f:
                sm_of.setdefault(f["ID"], f.get("SM"))
    cols = {k: [] for k in ("QNAME", "FLAG", "RNAME", "POS", "MAPQ", "CIGAR", "RNEXT", "PNEXT", "TLEN", "SEQ", "QUAL", "READ_GROUP_ID", "SAMPLE_ID")}
    tags = {n: [] for n, _, _ in std}
    aux_k, aux_v = [], []
    for l in text.split("\n"):
        if not l or l.startswith("@"):
            continue
        f = l.split("\t")
        cols["QNAME"].append(f[0].encode()); cols["FLAG"].append(int(f[1])); cols["RNAME"].append(f[2].encode()); cols["POS"].append(int(f[3]))
        cols["MAPQ"].append(int(f[4])); cols["CIGAR"].append(f[5].encode())
        cols["RNEXT"].append((f[2] if f[6] == "=" else f[6]).encode()); cols["PNEXT"].append(int(f[7])); cols["TLEN"].append(int(f[8]))
        cols["SEQ"].append(f[9].encode()); cols["QUAL"].append(f[10].encode())
        row = {}; ks, vs = [], []
        for t in f[11:]:
            name, ty, val = t.split(":", 2)
            if name in std_by and name not in row:
                row[name] = (ty, val)
            if name not in std_by:
                ks.append(name.encode()); vs.append(_aux_text(ty, val))
        rg = row.get("RG")
        cols["READ_GROUP_ID"].append(rg[1].encode() if rg and rg[0] == "Z" else None)
        sm = sm_of.get(rg[1]) if rg and rg[0] == "Z" else None
        cols["SAMPLE_ID"].append(sm.encode() if sm is not None else None)
        for n, (cty, _) in std_by.items():
            if n not in row:
                tags[n].append(None)
                continue
            ty, val = row[n]
            if cty == "i":
                tags[n].append(int(val) if ty == "i" else 0)          # bam_aux2i: 0 for a tag that is not an integer
            elif cty in "ZH":
                tags[n].append(val.encode() if ty in "ZH" else None)
            elif cty == "A":
                tags[n].append(val.encode() if ty == "A" else None)
            else:
                tags[n].append("skip")
        aux_k.append(ks or None); aux_v.append(vs or None)          # a row without such tags is NULL (bam_reader.c:982-991)
    return cols, tags, aux_k, aux_v


def check(table, std_cols, aux_cols, text, std):
    cols, tags, aux_k, aux_v = expectations(text, std)
    n = len(cols["QNAME"])
    assert table["n_rows"] == n
    for k, want in cols.items():
        got = [int(x) for x in table[k]] if k in ("FLAG", "POS", "MAPQ", "PNEXT", "TLEN") else list(table[k])
        assert got == want, (k, got[:4], want[:4])
    by = {c["name"]: orc.bcf_col_py(c) for c in std_cols}
    for name, want in tags.items():
        if any(w == "skip" for w in want):
            continue
        assert by[name] == want, (name, by[name][:4], want[:4])
    keys, vals = (orc.bcf_col_py(c) for c in aux_cols)
    assert keys == aux_k, (keys[:2], aux_k[:2])
    assert vals == aux_v, (vals[:2], aux_v[:2])


@pytest.mark.parametrize("name", FIXTURES)
def test_htslib_sam_fixture_oracle(name):
    import duckhts_amd
    text = _text(name)
    data = samtext.sam_text_as_bam(text)
    check(orc.bam_read(data), orc.bam_read_std_tags(data)["cols"], orc.bam_read_aux_map(data, exclude_standard=True)["cols"], text, duckhts_amd.std_tags())


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_htslib_sam_fixture_gpu(name):
    import duckhts_amd
    text = _text(name)
    for kw in ({}, {"level": 1, "payload": 777}):          # (one block, and the records cut into 777-byte BGZF blocks)
        data = samtext.sam_text_as_bam(text, **kw)
        got = duckhts_amd.read_bam(data, std_tags_cols=list(range(56)), aux_map="exclude_standard")
        check(got, got["tags"]["cols"], got["aux"]["cols"], text, duckhts_amd.std_tags())
        exp = orc.bam_read(data)
        for k in duckhts_amd.BAM_COLUMNS:
            assert list(got[k]) == list(exp[k]), k
