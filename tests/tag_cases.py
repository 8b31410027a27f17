"""BAM inputs exercising the standard-tag columns (row A5), shared by the CPU oracle tests and the GPU parity tests."""
import random
import struct

import bamwriter as bw

REFS = [("xx", 2000000), ("yy", 500000)]


def aux_tags_sam_equivalent():
    """test/data/aux_tags.sam.gz (the only record the reference's SQL test reads, duckhts.test:179-185) written as BAM"""
    recs = [bw.record("r1", 0, 0, 0, 60, "4M", seq="ACGT", qual="!!!!", tags=[("RG", "Z", "x1"), ("NM", "i", 2), ("XZ", "Z", "foo")])]
    return bw.bam_bytes([("xx", 20)], recs, text="@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:xx\tLN:20\n@RG\tID:x1\tSM:x1\n")


def type_matrix():
    recs = []
    T = lambda *t: list(t)
    # every integer width for 'i' columns, unsigned 32-bit beyond int32
    recs.append(bw.record("ints", 0, 0, 10, 60, "4M", seq="ACGT", qual="!!!!", tags=T(("AM", "c", -5), ("AS", "C", 250), ("CM", "s", -30000), ("CP", "S", 65000),
                                                                                         ("FI", "i", -2000000000), ("H0", "I", 4000000000), ("NM", "C", 7))))
    # 'i' columns holding non-integer types read as 0 (bam_aux2i), Z columns holding non-strings read NULL (bam_aux2Z)
    recs.append(bw.record("mismatch", 0, 0, 11, 60, "4M", seq="ACGT", qual="!!!!", tags=T(("NM", "Z", "oops"), ("AS", "f", 1.5), ("MD", "i", 5), ("RG", "A", "x"), ("TS", "Z", "++"),
                                                                                             ("SM", "B:C", [1, 2]), ("BC", "H", "1AE3"), ("UQ", "A", "7"))))
    # A column, A with NUL, B arrays of every subtype incl. float (doubles land in the BIGINT child), stored type != 'B'
    recs.append(bw.record("arrays", 0, 0, 12, 60, "4M", seq="ACGT", qual="!!!!", tags=T(("TS", "A", "+"), ("ML", "B:C", [0, 128, 255]), ("FZ", "B:S", [1, 65535]), ("MM", "Z", "C+m,5,12;"),
                                                                                           ("CG", "B:i", [-7, 9]))))
    recs.append(bw.record("arrays2", 0, 0, 13, 60, "4M", seq="ACGT", qual="!!!!", raw_aux=b"TSA\x00" + bw.aux_bytes([("ML", "B:f", [0.5, -2.25, 1e30]), ("FZ", "Z", "notB"), ("CG", "B:c", [-1, 2, -3])])
                          + b"MLBd" + struct.pack("<I", 1) + struct.pack("<d", 3.5)))
    # duplicates: first occurrence wins; empty strings; tags after a long list
    recs.append(bw.record("dups", 0, 0, 14, 60, "4M", seq="ACGT", qual="!!!!", tags=T(("NM", "i", 1), ("NM", "i", 2), ("LB", "Z", ""), ("PU", "Z", "unit"), ("ML", "B:C", list(range(200))), ("PG", "Z", "bwa"))))
    # no aux at all; unplaced read with tags
    recs.append(bw.record("none", 0, 0, 15, 60, "4M", seq="ACGT", qual="!!!!"))
    recs.append(bw.record("unplaced", 4, -1, -1, 0, "*", seq="ACGT", qual="!!!!", tags=T(("RG", "Z", "g"), ("OQ", "Z", "IIII"))))
    # corrupt aux: unknown type byte mid-way -> tags before it readable, at/after it NULL (bam_aux_get returns NULL)
    recs.append(bw.record("corrupt_mid", 0, 0, 16, 60, "4M", seq="ACGT", qual="!!!!", raw_aux=bw.aux_bytes([("NM", "i", 3), ("MD", "Z", "4")]) + b"XX?" + b"\x01\x02" + bw.aux_bytes([("AS", "i", 9)])))
    # Z tag without terminator at the end of the record
    recs.append(bw.record("unterminated", 0, 0, 17, 60, "4M", seq="ACGT", qual="!!!!", raw_aux=bw.aux_bytes([("NM", "i", 4)]) + b"PGZabc"))
    # B array whose count runs past the record
    recs.append(bw.record("b_overrun", 0, 0, 18, 60, "4M", seq="ACGT", qual="!!!!", raw_aux=bw.aux_bytes([("AS", "i", 5)]) + b"MLBC" + struct.pack("<I", 1000) + b"\x01\x02" ))
    # long CIGAR swap: the CG tag is consumed by bam_read1 and must not show up in the CG column; tags around it still do
    ops = [(0, 1)] * 70000
    real = b"".join(struct.pack("<I", (l << 4) | op) for op, l in ops)
    aux = bw.aux_bytes([("NM", "i", 6)]) + b"CGBI" + struct.pack("<I", len(ops)) + real + bw.aux_bytes([("MD", "Z", "70000"), ("RG", "Z", "g")])
    recs.append(bw.record("longcig", 0, 0, 100, 60, raw_cigar=[(70000 << 4) | 4, (70000 << 4) | 3], seq="A" * 70000, qual=b"\x1e" * 70000, raw_aux=aux))
    return bw.bam_bytes(REFS, recs)


def fuzz(seed=3, n=4000, payload=3000):
    rnd = random.Random(seed)
    names = ["AM", "AS", "BC", "CG", "FZ", "ML", "MD", "NM", "RG", "TS", "SA", "MM", "XX", "YY", "UQ", "OQ"]
    recs = []
    for i in range(n):
        tags = []
        for nm in rnd.sample(names, rnd.randint(0, 9)):
            k = rnd.random()
            if k < 0.35:
                tags.append((nm, rnd.choice("cCsSiI"), {"c": rnd.randint(-128, 127), "C": rnd.randint(0, 255), "s": rnd.randint(-32768, 32767), "S": rnd.randint(0, 65535),
                                                        "i": rnd.randint(-2**31, 2**31 - 1), "I": rnd.randint(0, 2**32 - 1)}))
                t = tags[-1]
                tags[-1] = (t[0], t[1], t[2][t[1]])
            elif k < 0.6:
                tags.append((nm, rnd.choice("ZH"), "".join(rnd.choices("ACGT0123,;+", k=rnd.randint(0, 30)))))
            elif k < 0.7:
                tags.append((nm, "A", rnd.choice("+-xyz")))
            elif k < 0.8:
                tags.append((nm, "f", rnd.uniform(-10, 10)))
            else:
                sub = rnd.choice("cCsSiIf")
                lim = {"c": (-128, 127), "C": (0, 255), "s": (-32768, 32767), "S": (0, 65535), "i": (-2**31, 2**31 - 1), "I": (0, 2**32 - 1)}
                vals = [rnd.uniform(-5, 5) if sub == "f" else rnd.randint(*lim[sub]) for _ in range(rnd.randint(0, 12))]
                tags.append((nm, "B:" + sub, vals))
        recs.append(bw.record("q%d" % i, 0, rnd.randrange(2), i * 3, 60, "10M", seq="ACGTACGTAC", qual="IIIIIIIIII", tags=tags))
    return bw.bam_bytes(REFS, recs, payload=payload)
