"""Minimal BCF2.2 writer for edge-case fixtures (test infrastructure).

Layout per the BCF2 specification as consumed by bcf_read1_core / bcf_record_check (htslib vcf.c:1874-1911, 2040-2212):
  "BCF\\2\\2", u32 l_text, text (NUL-terminated), then records
  [u32 l_shared][u32 l_indiv][i32 CHROM][i32 POS][i32 rlen][f32 QUAL][u16 n_info][u16 n_allele][u24 n_sample][u8 n_fmt] shared indiv
"""
import struct

from bamwriter import bgzf_file

INT8_MISSING, INT8_END = -128, -127
INT16_MISSING, INT16_END = -32768, -32767
INT32_MISSING, INT32_END = -2147483648, -2147483647
FLOAT_MISSING, FLOAT_END = 0x7F800001, 0x7F800002
BT_NULL, BT_INT8, BT_INT16, BT_INT32, BT_FLOAT, BT_CHAR = 0, 1, 2, 3, 5, 7

MISSING = "missing"      # sentinel objects for value lists
END = "end"


def typed_int1(v: int, width=None) -> bytes:
    if width is None:
        width = 1 if -120 <= v <= 127 else 2 if -32000 <= v <= 32767 else 4
    if width == 1:
        return b"\x11" + struct.pack("<b", v)
    if width == 2:
        return b"\x12" + struct.pack("<h", v)
    return b"\x13" + struct.pack("<i", v)


def desc(n: int, t: int) -> bytes:
    if n < 15:
        return bytes([(n << 4) | t])
    return bytes([0xF0 | t]) + typed_int1(n)


def tv_str(s: bytes) -> bytes:
    return desc(len(s), BT_CHAR) + s


def _int_width(vals):
    w = 1
    for v in vals:
        if isinstance(v, str):
            continue
        if not -120 <= v <= 127:
            w = max(w, 2)
        if not -32000 <= v <= 32767:
            w = 4
    return w


def pack_ints(vals, width) -> bytes:
    fmt = {1: "<b", 2: "<h", 4: "<i"}[width]
    miss = {1: INT8_MISSING, 2: INT16_MISSING, 4: INT32_MISSING}[width]
    end = {1: INT8_END, 2: INT16_END, 4: INT32_END}[width]
    return b"".join(struct.pack(fmt, miss if v == MISSING else end if v == END else v) for v in vals)


def pack_floats(vals) -> bytes:
    out = b""
    for v in vals:
        if v == MISSING:
            out += struct.pack("<I", FLOAT_MISSING)
        elif v == END:
            out += struct.pack("<I", FLOAT_END)
        elif isinstance(v, int) and not isinstance(v, bool) and v > 0xFFFF:
            out += struct.pack("<I", v)          # raw bit pattern
        else:
            out += struct.pack("<f", v)
    return out


def tv_ints(vals, width=None) -> bytes:
    width = width or _int_width(vals)
    return desc(len(vals), {1: BT_INT8, 2: BT_INT16, 4: BT_INT32}[width]) + pack_ints(vals, width)


def tv_floats(vals) -> bytes:
    return desc(len(vals), BT_FLOAT) + pack_floats(vals)


def gt(*alleles, phased=False):
    """GT integers: (allele+1)<<1 | phased; None = missing allele (0)."""
    return [0 if a is None else ((a + 1) << 1) | (1 if phased else 0) for a in alleles]


def fmt_ints(key, per_sample, width=None, n=None) -> bytes:
    """per_sample: list of value lists; shorter lists are padded with END."""
    n = n if n is not None else max((len(v) for v in per_sample), default=0)
    flat = [x for v in per_sample for x in (list(v) + [END] * (n - len(v)))]
    width = width or _int_width(flat)
    return typed_int1(key) + desc(n, {1: BT_INT8, 2: BT_INT16, 4: BT_INT32}[width]) + pack_ints(flat, width)


def fmt_floats(key, per_sample, n=None) -> bytes:
    n = n if n is not None else max((len(v) for v in per_sample), default=0)
    flat = [x for v in per_sample for x in (list(v) + [END] * (n - len(v)))]
    return typed_int1(key) + desc(n, BT_FLOAT) + pack_floats(flat)


def fmt_strs(key, per_sample) -> bytes:
    n = max((len(v) for v in per_sample), default=0)
    return typed_int1(key) + desc(n, BT_CHAR) + b"".join(v + b"\0" * (n - len(v)) for v in per_sample)


def record(rid=0, pos=0, rlen=1, qual=None, id=b"", alleles=(b"A",), filters=None, info=(), fmt=(), n_sample=0,
           n_allele=None, n_info=None, n_fmt=None, qual_bits=None, filter_raw=None) -> bytes:
    """info: list of (key, typed_value_bytes) ; fmt: list of bytes from fmt_* ; filters: list of dictionary ids or None (= empty vector)."""
    shared = tv_str(id)
    for a in alleles:
        shared += tv_str(a)
    if filter_raw is not None:
        shared += filter_raw
    elif not filters:
        shared += b"\x00"
    else:
        shared += tv_ints(list(filters))
    for key, tv in info:
        shared += typed_int1(key) + tv
    indiv = b"".join(fmt)
    qb = qual_bits if qual_bits is not None else (FLOAT_MISSING if qual is None else struct.unpack("<I", struct.pack("<f", qual))[0])
    core = struct.pack("<iiiIHHI", rid, pos, rlen, qb, len(info) if n_info is None else n_info,
                       len(alleles) if n_allele is None else n_allele,
                       (n_sample & 0xFFFFFF) | ((len(fmt) if n_fmt is None else n_fmt) << 24))
    return struct.pack("<II", 24 + len(shared), len(indiv)) + core + shared + indiv


def header(lines, samples=(), contigs=("1", "2"), fileformat="VCFv4.2") -> str:
    out = [f"##fileformat={fileformat}"] if fileformat else []
    out += [f"##contig=<ID={c}>" if isinstance(c, str) else c[0] for c in contigs]
    out += list(lines)
    cols = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"
    if samples:
        cols += "\tFORMAT\t" + "\t".join(samples)
    return "\n".join(out + [cols]) + "\n"


def bcf_raw(header_text: str, records) -> bytes:
    t = header_text.encode() + b"\0"
    return b"BCF\x02\x02" + struct.pack("<I", len(t)) + t + b"".join(records)


def bcf_bytes(header_text: str, records, **kw) -> bytes:
    return bgzf_file(bcf_raw(header_text, records), **kw)
