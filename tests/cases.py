"""Shared BAM edge-case inputs (bytes) used by both the CPU (oracle) and GPU parity tests."""
import random
import struct
import zlib

import bamwriter as bw

REFS = [("chr1", 100000), ("chr2", 50000), ("chrM", 16569)]
TEXT = ("@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:100000\n@SQ\tSN:chr2\tLN:50000\n@SQ\tSN:chrM\tLN:16569\n"
        "@RG\tID:g1\tSM:sampleA\tPL:X\n@RG\tID:g2\tPL:X\n@RG\tID:g1\tSM:dup\n@RG\tID:g3\tSM:\tPL:q\n@CO\tfree text\n")


def _rand_seq(rng, n):
    return "".join(rng.choice("ACGTN=MRSVWYHKDB") for _ in range(n))


def basic_records(rng, n=300):
    recs = []
    for i in range(n):
        l = rng.choice([0, 1, 2, 15, 16, 17, 31, 32, 33, 100, 150, 151, 255, 256, 257, 300])
        seq = _rand_seq(rng, l) if l else "*"
        kind = rng.randrange(8)
        if l == 0:
            cigar = rng.choice(["*", "10M", "5S"])
        elif kind == 0:
            cigar = "*"
        elif kind == 1 and l > 4:
            a = rng.randrange(1, l - 1)
            cigar = f"{a}S{l - a}M"
        elif kind == 2 and l > 6:
            a = rng.randrange(1, l - 3)
            cigar = f"{a}M2I{l - a - 2}M" if l - a - 2 > 0 else f"{l}M"
        elif kind == 3 and l > 6:
            a = rng.randrange(1, l - 1)
            cigar = f"{a}M1000000N{l - a}M3H"
        elif kind == 4:
            cigar = f"{l}="
        else:
            cigar = f"{l}M"
        qual = None if rng.random() < 0.15 else "".join(chr(33 + rng.randrange(0, 60)) for _ in range(l))
        tags = []
        if rng.random() < 0.7:
            tags.append(("NM", "C", rng.randrange(200)))
        if rng.random() < 0.5:
            tags.append(("MD", "Z", str(l)))
        if rng.random() < 0.3:
            tags.append(("XB", "B:s", [rng.randrange(-300, 300) for _ in range(rng.randrange(0, 6))]))
        r = rng.random()
        if r < 0.5:
            tags.insert(rng.randrange(len(tags) + 1), ("RG", "Z", rng.choice(["g1", "g2", "g3", "zz", ""])))
        elif r < 0.55:
            tags.append(("RG", "i", 7))          # wrong type: bam_aux2Z -> NULL
        if rng.random() < 0.3:
            tags.append(("XF", "f", 1.5))
        if rng.random() < 0.2:
            tags.append(("XA", "A", "q"))
        tid = rng.choice([-1, 0, 1, 2])
        recs.append(bw.record(qname=f"read{i}:{rng.randrange(10 ** rng.randrange(1, 12))}", flag=rng.choice([0, 4, 16, 99, 147, 83, 163, 1024, 2048, 65535]),
                              tid=tid, pos=rng.randrange(-1, 90000), mapq=rng.randrange(256), cigar=cigar,
                              mtid=rng.choice([-1, 0, 1, 2]), mpos=rng.randrange(-1, 90000), tlen=rng.randrange(-5000, 5000),
                              seq=seq, qual=qual, tags=tags))
    return recs


def case_basic(payload=65280, level=6, seed=1, n=300, **kw):
    rng = random.Random(seed)
    return bw.bam_bytes(REFS, basic_records(rng, n), text=TEXT, payload=payload, level=level, **kw)


def case_quirks():
    """QUAL byte 223 (+33 == NUL) truncation, qual[0]==0xFF, QNAME without NUL / with embedded NUL, long QNAME,
    unknown CIGAR ops, CG:B,I long-CIGAR swap, unmapped with CIGAR/qlen mismatch tolerated."""
    r = []
    r.append(bw.record(qname="q223", cigar="6M", seq="ACGTAC", raw_qual=bytes([10, 20, 223, 30, 40, 50])))
    r.append(bw.record(qname="q223first", cigar="4M", seq="ACGT", raw_qual=bytes([223, 1, 2, 3])))
    r.append(bw.record(qname="qff", cigar="4M", seq="ACGT", raw_qual=bytes([255, 1, 2, 3])))
    r.append(bw.record(qname="qff2", cigar="4M", seq="ACGT", raw_qual=bytes([1, 255, 2, 3])))
    r.append(bw.record(raw_qname=b"nonul", cigar="4M", seq="ACGT"))
    r.append(bw.record(raw_qname=b"emb\x00edded\x00", cigar="4M", seq="ACGT"))
    r.append(bw.record(raw_qname=b"\x00", cigar="4M", seq="ACGT"))
    r.append(bw.record(qname="x" * 254, cigar="4M", seq="ACGT"))
    r.append(bw.record(qname="ops", raw_cigar=[(3 << 4) | 10, (1 << 4) | 15, (4 << 4) | 9], seq="ACGT", flag=4))
    # CG swap: fake CIGAR kS + real CIGAR in CG:B,I (sam.c:675-730); RG after CG must still be found
    real = [(2 << 4) | 0, (1 << 4) | 1, (3 << 4) | 0]
    r.append(bw.record(qname="cgswap", tid=0, pos=10, raw_cigar=[(6 << 4) | 4, (6 << 4) | 3], seq="ACGTAC",
                       tags=[("NM", "C", 1), ("CG", "B:I", real), ("RG", "Z", "g1")]))
    r.append(bw.record(qname="cgswap_first", tid=0, pos=10, raw_cigar=[(6 << 4) | 4], seq="ACGTAC",
                       tags=[("CG", "B:I", real), ("RG", "Z", "g2")]))
    r.append(bw.record(qname="cg_unplaced", tid=-1, pos=-1, raw_cigar=[(6 << 4) | 4], seq="ACGTAC", flag=4,
                       tags=[("CG", "B:I", real), ("RG", "Z", "g1")]))
    r.append(bw.record(qname="unmapped_mismatch", flag=4, cigar="3M", seq="ACGTAC"))
    r.append(bw.record(qname="noseq_cigar", cigar="10M", seq="*"))
    r.append(bw.record(qname="big", cigar="1M1I" * 3000, seq="A" * 6000, qual="Q" * 6000, tags=[("RG", "Z", "g1")]))
    r.append(bw.record(qname="huge_oplen", cigar="268435455M", seq="*"))
    r.append(bw.record(qname="last", cigar="4M", seq="ACGT", tags=[("RG", "H", "1AE3")]))
    return bw.bam_bytes(REFS, r, text=TEXT, payload=700, level=6)


def case_long_record():
    """htslib test.pl:852-864: 1M1I x 16000, 32 kb SEQ; record spans several BGZF blocks and many tiles."""
    r = [bw.record(qname="pre", cigar="4M", seq="ACGT"),
         bw.record(qname="read", flag=0, tid=0, pos=0, mapq=60, cigar="1M1I" * 16000, seq="A" * 32000, qual="Q" * 32000),
         bw.record(qname="post", cigar="4M", seq="ACGT")]
    return bw.bam_bytes([("ref", 100000)], r, text="@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ref\tLN:100000\n", payload=30000, level=0)


def case_error_midfile(kind):
    rng = random.Random(7)
    recs = basic_records(rng, 120)
    bad = {
        "blocklen_small": struct.pack("<i", 20) + b"\x00" * 20,
        "inconsistent": bw.record(qname="bad", cigar="4M", seq="ACGT")[:4] + struct.pack("<iiIIiiii", 0, 0, 4, 500, 4, -1, -1, 0) + b"bad\x00ACGT" + b"\x00" * 10,
        "cigar_qlen": bw.record(qname="badcig", cigar="3M", seq="ACGTAC"),
        "tid_range": bw.record(qname="badtid", tid=3, cigar="4M", seq="ACGT"),
        "mtid_range": bw.record(qname="badmtid", mtid=-2, cigar="4M", seq="ACGT"),
        "neg_lseq": bw.record(qname="neg", cigar="*", seq="*", l_seq=-5),
        "lq0": bw.record(raw_qname=b"", cigar="*", seq="*"),
    }[kind]
    if kind == "inconsistent":
        bad = struct.pack("<i", len(bad) - 4) + bad[4:]
    recs = recs[:77] + [bad] + recs[77:]
    return bw.bam_bytes(REFS, recs, text=TEXT, payload=1500)


def case_truncated():
    data = case_basic(payload=4000, n=150)
    return data[: len(data) - 28 - 300]            # cuts the last data block short, no EOF block


def case_truncated_record():
    rng = random.Random(3)
    raw = b"".join(basic_records(rng, 60))
    raw = raw[: len(raw) - 17]                      # last record incomplete, blocks themselves intact
    return bw.bgzf_file(bw.bam_header(REFS, TEXT), eof=False) + bw.bgzf_file(raw, payload=3000)


def case_empty_blocks():
    rng = random.Random(5)
    raw = b"".join(basic_records(rng, 80))
    hdr = bw.bgzf_file(bw.bam_header(REFS, TEXT), eof=False)
    mid = len(raw) // 2
    empty = bw.bgzf_block(b"")
    return hdr + empty + bw.bgzf_file(raw[:mid], payload=2000, eof=False) + empty + empty + bw.bgzf_file(raw[mid:], payload=2500)


def case_bad_crc():
    data = bytearray(case_basic(payload=3000, n=120))
    # flip one CRC byte in the 5th block
    pos, k = 0, 0
    while True:
        bl = struct.unpack_from("<H", data, pos + 16)[0] + 1
        if k == 4:
            data[pos + bl - 8] ^= 0x55
            break
        pos += bl
        k += 1
    return bytes(data)


def case_bad_deflate():
    data = bytearray(case_basic(payload=3000, n=120))
    pos, k = 0, 0
    while True:
        bl = struct.unpack_from("<H", data, pos + 16)[0] + 1
        if k == 6:
            for j in range(30, 60):
                data[pos + j] ^= 0xA5
            break
        pos += bl
        k += 1
    return bytes(data)


def case_header_only():
    return bw.bgzf_file(bw.bam_header(REFS, TEXT))


def case_no_refs_unmapped():
    r = [bw.record(qname=f"u{i}", flag=4, tid=-1, pos=-1, seq="ACGT", qual="IIII") for i in range(10)]
    return bw.bam_bytes([], r, text="@HD\tVN:1.6\n")


def case_bad_header_text():
    """malformed @RG line: the whole dictionary is void, SAMPLE_ID NULL everywhere"""
    r = [bw.record(qname="a", cigar="4M", seq="ACGT", tags=[("RG", "Z", "g1")])]
    return bw.bam_bytes(REFS, r, text="@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:100000\n@RG\tID:g1\tSM:s\n@RG\tbroken\n")


def case_fixed_huffman():
    rng = random.Random(11)
    return bw.bam_bytes(REFS, basic_records(rng, 40), text=TEXT, payload=900, level=6, strategy=zlib.Z_FIXED)


ALL_CASES = {
    "basic": case_basic,
    "basic_small_blocks": lambda: case_basic(payload=777, n=200, seed=2),
    "basic_tiny_blocks": lambda: case_basic(payload=61, n=60, seed=3),
    "basic_stored": lambda: case_basic(level=0, n=200, seed=4),
    "basic_level1": lambda: case_basic(level=1, n=200, seed=5),
    "basic_level9": lambda: case_basic(level=9, n=200, seed=6),
    "fixed_huffman": case_fixed_huffman,
    "quirks": case_quirks,
    "long_record": case_long_record,
    "err_blocklen_small": lambda: case_error_midfile("blocklen_small"),
    "err_inconsistent": lambda: case_error_midfile("inconsistent"),
    "err_cigar_qlen": lambda: case_error_midfile("cigar_qlen"),
    "err_tid_range": lambda: case_error_midfile("tid_range"),
    "err_mtid_range": lambda: case_error_midfile("mtid_range"),
    "err_neg_lseq": lambda: case_error_midfile("neg_lseq"),
    "err_lq0": lambda: case_error_midfile("lq0"),
    "truncated_file": case_truncated,
    "truncated_record": case_truncated_record,
    "empty_blocks": case_empty_blocks,
    "bad_crc": case_bad_crc,
    "bad_deflate": case_bad_deflate,
    "header_only": case_header_only,
    "no_refs_unmapped": case_no_refs_unmapped,
    "bad_header_text": case_bad_header_text,
}
