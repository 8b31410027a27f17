"""ctypes binding of oracle/liborc.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class StrCol(C.Structure):
    _fields_ = [("off", C.POINTER(C.c_uint64)), ("bytes", C.POINTER(C.c_uint8)), ("valid", C.POINTER(C.c_uint8)),
                ("n", C.c_size_t), ("cap_n", C.c_size_t), ("nbytes", C.c_size_t), ("cap_bytes", C.c_size_t)]

    def to_list(self):
        n = self.n
        if n == 0:
            return []
        off = np.ctypeslib.as_array(self.off, (n + 1,))
        data = bytes(np.ctypeslib.as_array(self.bytes, (max(self.nbytes, 1),))[: self.nbytes])
        valid = np.ctypeslib.as_array(self.valid, (n,))
        return [data[off[i]:off[i + 1]] if valid[i] else None for i in range(n)]


class Bgzf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("n_blocks", C.c_int64), ("status", C.c_int),
                ("has_eof_marker", C.c_int), ("coff", C.POINTER(C.c_int64)), ("clen", C.POINTER(C.c_int32)),
                ("ulen", C.POINTER(C.c_int32))]


class Bam(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("status", C.c_int), ("n_ref", C.c_int32),
                ("flag", C.POINTER(C.c_uint16)), ("pos", C.POINTER(C.c_int64)), ("mapq", C.POINTER(C.c_int32)),
                ("pnext", C.POINTER(C.c_int64)), ("tlen", C.POINTER(C.c_int64)),
                ("tid", C.POINTER(C.c_int32)), ("mtid", C.POINTER(C.c_int32)), ("rec_off", C.POINTER(C.c_int64)),
                ("qname", StrCol), ("rname", StrCol), ("cigar", StrCol), ("rnext", StrCol), ("seq", StrCol),
                ("qual", StrCol), ("rg", StrCol), ("sample", StrCol), ("ref_names", StrCol),
                ("ref_len", C.POINTER(C.c_int32)), ("text", C.c_char_p), ("l_text", C.c_size_t),
                ("first_rec_off", C.c_int64)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liborc.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_inflate_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_inflate_raw.restype = C.c_int
        L.orc_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.orc_crc32.restype = C.c_uint32
        L.orc_bgzf_inflate_all.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Bgzf)]
        L.orc_bgzf_inflate_all.restype = C.c_int
        L.orc_bgzf_free.argtypes = [C.POINTER(Bgzf)]
        L.orc_bam_read.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Bam)]
        L.orc_bam_read.restype = C.c_int
        L.orc_bam_free.argtypes = [C.POINTER(Bam)]
        L.orc_bam_scan_count.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
        L.orc_bam_scan_count.restype = C.c_int64
        L.orc_use_system_zlib.argtypes = [C.c_int]
        L.orc_use_system_zlib.restype = C.c_int
        _LIB = L
    return _LIB


def inflate_raw(src: bytes, cap: int = 65536):
    out = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    r = lib().orc_inflate_raw(src, len(src), out, cap, C.byref(n))
    return r, out.raw[: n.value]


def crc32(data: bytes, crc: int = 0) -> int:
    return lib().orc_crc32(crc, data, len(data))


def bgzf_inflate_all(file_bytes: bytes):
    b = Bgzf()
    lib().orc_bgzf_inflate_all(file_bytes, len(file_bytes), C.byref(b))
    nb = b.n_blocks
    res = {
        "status": b.status, "n_blocks": nb, "has_eof": b.has_eof_marker,
        "data": bytes(np.ctypeslib.as_array(b.data, (max(b.len, 1),))[: b.len]),
        "coff": np.ctypeslib.as_array(b.coff, (nb,)).copy() if nb else np.zeros(0, np.int64),
        "clen": np.ctypeslib.as_array(b.clen, (nb,)).copy() if nb else np.zeros(0, np.int32),
        "ulen": np.ctypeslib.as_array(b.ulen, (nb,)).copy() if nb else np.zeros(0, np.int32),
    }
    lib().orc_bgzf_free(C.byref(b))
    return res


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, (n,)).astype(dtype, copy=True)


def bam_read(file_bytes: bytes):
    """Returns a dict of the 13 read_bam core columns (+ dictionary ids, record offsets)."""
    b = Bam()
    lib().orc_bam_read(file_bytes, len(file_bytes), C.byref(b))
    n = b.n_rows
    res = {
        "n_rows": n, "status": b.status, "n_ref": b.n_ref,
        "FLAG": _arr(b.flag, n, np.uint16), "POS": _arr(b.pos, n, np.int64), "MAPQ": _arr(b.mapq, n, np.int32),
        "PNEXT": _arr(b.pnext, n, np.int64), "TLEN": _arr(b.tlen, n, np.int64),
        "tid": _arr(b.tid, n, np.int32), "mtid": _arr(b.mtid, n, np.int32), "rec_off": _arr(b.rec_off, n, np.int64),
        "QNAME": b.qname.to_list(), "RNAME": b.rname.to_list(), "CIGAR": b.cigar.to_list(),
        "RNEXT": b.rnext.to_list(), "SEQ": b.seq.to_list(), "QUAL": b.qual.to_list(),
        "READ_GROUP_ID": b.rg.to_list(), "SAMPLE_ID": b.sample.to_list(),
        "ref_names": b.ref_names.to_list(), "ref_len": _arr(b.ref_len, b.n_ref, np.int32),
        "text": (b.text[: b.l_text] if b.text else b""), "first_rec_off": b.first_rec_off,
    }
    lib().orc_bam_free(C.byref(b))
    return res


def bam_scan_count(file_bytes: bytes):
    st = C.c_int(0)
    n = lib().orc_bam_scan_count(file_bytes, len(file_bytes), C.byref(st))
    return n, st.value


def use_system_zlib(on: bool) -> bool:
    """timing leg only (bench.py cpu_baseline): route inflate/crc32 through libz.so.1 like the reference does"""
    return bool(lib().orc_use_system_zlib(int(on)))
