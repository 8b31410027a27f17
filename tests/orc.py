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


def bam_digest(file_bytes):
    """CRC-32 digests of the oracle's 13 columns (layout: oracle/dhts_oracle.c orc_bam_digest) -> uint32[32]"""
    out = (C.c_uint32 * 32)()
    L = lib()
    L.orc_bam_digest.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
    L.orc_bam_digest(file_bytes, len(file_bytes), out)
    return [int(x) for x in out]


def bam_scan_count_mt(file_bytes, n_threads: int):
    """timing leg only (bench.py cpu_baseline): n_threads inflate workers + one scan thread, the shape of the reference's read path"""
    st = C.c_int(0)
    L = lib()
    L.orc_bam_scan_count_mt.restype = C.c_int64
    L.orc_bam_scan_count_mt.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int)]
    n = L.orc_bam_scan_count_mt(file_bytes, len(file_bytes), n_threads, C.byref(st))
    return n, st.value


def use_system_zlib(on: bool) -> bool:
    """timing leg only (bench.py cpu_baseline): route inflate/crc32 through libz.so.1 like the reference does"""
    return bool(lib().orc_use_system_zlib(int(on)))


# ---- read_bcf ---------------------------------------------------------------------------------------------------
BCF_TYPES = {1: "VARCHAR", 2: "BIGINT", 3: "DOUBLE", 4: "BOOLEAN", 5: "INTEGER", 6: "FLOAT"}


def decode_bcf_blob(blob: bytes):
    """Canonical column blob (oracle/bcf_oracle.c, also produced by duckhts_amd.read_bcf) -> dict."""
    mv = memoryview(blob)
    ncol, = np.frombuffer(mv[0:4], np.uint32)
    nrows, = np.frombuffer(mv[4:12], np.uint64)
    status, = np.frombuffer(mv[12:16], np.int32)
    first, = np.frombuffer(mv[16:24], np.uint64)
    nsmp, = np.frombuffer(mv[24:28], np.uint32)
    n = int(nrows)
    pos = 28
    cols = []

    def take(cnt, dtype):
        nonlocal pos
        sz = cnt * np.dtype(dtype).itemsize
        a = np.frombuffer(mv[pos:pos + sz], dtype)
        pos += sz
        return a

    def payload(c, cnt, pre):
        if c["type"] == 1:
            c[pre + "soff"] = take(cnt + 1, np.uint64)
            c[pre + "sbytes"] = take(int(c[pre + "soff"][-1]), np.uint8)
        else:
            c[pre + "fixed"] = take(cnt, np.uint64)

    for _ in range(int(ncol)):
        nl = int(take(1, np.uint16)[0])
        name = bytes(take(nl, np.uint8)).decode()
        t, il = (int(x) for x in take(2, np.uint8))
        c = {"name": name, "type": t, "is_list": 1 if il else 0, "valid": take(n, np.uint8)}
        if not il:
            payload(c, n, "")
        else:
            le = take(2 * n, np.uint64).reshape(n, 2)
            c["loff"], c["llen"] = le[:, 0].copy(), le[:, 1].copy()
            c["child_n"] = int(take(1, np.uint64)[0])
            payload(c, c["child_n"], "c")
            if il == 2:                                   # list with NULL elements (VEP_* columns)
                c["cvalid"] = take(c["child_n"], np.uint8)
        cols.append(c)
    nrec = int(take(1, np.uint64)[0])
    rec = {"rid": take(nrec, np.int64).copy(), "pos0": take(nrec, np.int64).copy(), "rlen": take(nrec, np.int64).copy()}
    assert pos == len(blob), (pos, len(blob))
    return {"n_rows": n, "rec": rec, "status": int(status), "first_rec_off": int(first), "n_samples": int(nsmp), "cols": cols,
            "by_name": {c["name"]: c for c in cols}}


def bcf_read(file_bytes: bytes, tidy: bool = False):
    L = lib()
    L.orc_bcf_read.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
    L.orc_bcf_read.restype = C.c_int
    L.orc_free.argtypes = [C.c_void_p]
    blob = C.c_void_p()
    n = C.c_size_t(0)
    rows = C.c_int64(0)
    st = L.orc_bcf_read(file_bytes, len(file_bytes), int(tidy), 1, C.byref(blob), C.byref(n), C.byref(rows))
    if st <= -100:
        return {"status": st, "n_rows": 0, "cols": [], "by_name": {}}
    data = C.string_at(blob, n.value)
    L.orc_free(blob)
    res = decode_bcf_blob(data)
    assert res["status"] == st
    return res


def bcf_scan_count(file_bytes: bytes, tidy: bool = False):
    L = lib()
    L.orc_bcf_read.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
    L.orc_bcf_read.restype = C.c_int
    rows = C.c_int64(0)
    st = L.orc_bcf_read(file_bytes, len(file_bytes), int(tidy), 0, None, None, C.byref(rows))
    return rows.value, st


def _scalar_py(c, i, pre=""):
    t = c["type"]
    if t == 1:
        return bytes(c[pre + "sbytes"][int(c[pre + "soff"][i]):int(c[pre + "soff"][i + 1])])
    b = int(c[pre + "fixed"][i])
    if t == 2:
        return b - (1 << 64) if b >> 63 else b
    if t == 3:
        return float(np.array([b], np.uint64).view(np.float64)[0])
    if t == 4:
        return bool(b)
    if t == 5:
        b &= 0xFFFFFFFF
        return b - (1 << 32) if b >> 31 else b
    return float(np.array([b & 0xFFFFFFFF], np.uint32).view(np.float32)[0])


def bcf_col_py(c):
    """Column -> list of python values (None = NULL, lists for LIST columns); for small fixtures."""
    out = []
    for i in range(len(c["valid"])):
        if not c["valid"][i]:
            out.append(None)
        elif c["is_list"]:
            o, l = int(c["loff"][i]), int(c["llen"][i])
            out.append([_scalar_py(c, j, "c") if ("cvalid" not in c or c["cvalid"][j]) else None for j in range(o, o + l)])
        else:
            out.append(_scalar_py(c, i))
    return out


def bcf_cols_diff(a, b):
    """First difference between two decoded tables (None if identical)."""
    if a["n_rows"] != b["n_rows"]:
        return f"n_rows {a['n_rows']} != {b['n_rows']}"
    if len(a["cols"]) != len(b["cols"]):
        return f"ncol {len(a['cols'])} != {len(b['cols'])}"
    for ca, cb in zip(a["cols"], b["cols"]):
        for k in ("name", "type", "is_list"):
            if ca[k] != cb[k]:
                return f"{ca['name']}: {k} {ca[k]} != {cb[k]}"
        for k in ("valid", "fixed", "soff", "sbytes", "loff", "llen", "cfixed", "csoff", "csbytes", "cvalid"):
            if (k in ca) != (k in cb):
                return f"{ca['name']}: field {k} presence"
            if k in ca and not np.array_equal(ca[k], cb[k]):
                bad = np.nonzero(ca[k][:min(len(ca[k]), len(cb[k]))] != cb[k][:min(len(ca[k]), len(cb[k]))])[0]
                return f"{ca['name']}: {k} differs (len {len(ca[k])} vs {len(cb[k])}, first at {bad[0] if len(bad) else 'tail'})"
    return None


def bcf_take_rows(table, rows):
    """Row subset (in the given order, repeats allowed) of a decoded table -> new table in the same canonical layout."""
    rows = np.asarray(rows, np.int64)
    cols = []
    for c in table["cols"]:
        o = {"name": c["name"], "type": c["type"], "is_list": c["is_list"], "valid": c["valid"][rows]}

        def gather(off, data, idx):
            ln = (off[1:] - off[:-1])[idx].astype(np.int64)
            noff = np.concatenate([[0], np.cumsum(ln)]).astype(np.uint64)
            tot = int(noff[-1])
            src = np.repeat(off[:-1][idx].astype(np.int64) - noff[:-1].astype(np.int64), ln) + np.arange(tot) if tot else np.zeros(0, np.int64)
            return noff, data[src]
        if not c["is_list"]:
            if "fixed" in c:
                o["fixed"] = c["fixed"][rows]
            else:
                o["soff"], o["sbytes"] = gather(c["soff"], c["sbytes"], rows)
        else:
            ln = c["llen"][rows].astype(np.int64)
            noff = np.concatenate([[0], np.cumsum(ln)]).astype(np.uint64)
            o["loff"], o["llen"], o["child_n"] = noff[:-1].copy(), c["llen"][rows], int(noff[-1])
            tot = int(noff[-1])
            kid = np.repeat(c["loff"][rows].astype(np.int64) - noff[:-1].astype(np.int64), ln) + np.arange(tot) if tot else np.zeros(0, np.int64)
            if "cfixed" in c:
                o["cfixed"] = c["cfixed"][kid]
            else:
                o["csoff"], o["csbytes"] = gather(c["csoff"], c["csbytes"], kid)
            if "cvalid" in c:
                o["cvalid"] = c["cvalid"][kid]
        cols.append(o)
    return {"n_rows": len(rows), "cols": cols, "by_name": {c["name"]: c for c in cols}}


def bam_read_std_tags(file_bytes: bytes):
    """the 56 standard-tag columns (read_bam(standard_tags := true)) as a decoded canonical table"""
    L = lib()
    L.orc_bam_read_std_tags.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.orc_bam_read_std_tags.restype = C.c_int
    L.orc_free.argtypes = [C.c_void_p]
    blob, n = C.c_void_p(), C.c_size_t(0)
    L.orc_bam_read_std_tags(file_bytes, len(file_bytes), C.byref(blob), C.byref(n))
    data = C.string_at(blob, n.value)
    L.orc_free(blob)
    return decode_bcf_blob(data)


def bam_read_aux_map(file_bytes: bytes, exclude_standard: bool = True):
    """AUXILIARY_TAGS as two LIST(VARCHAR) columns (keys, rendered values) in the canonical table layout"""
    L = lib()
    L.orc_bam_read_aux_map.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.orc_bam_read_aux_map.restype = C.c_int
    L.orc_free.argtypes = [C.c_void_p]
    blob, n = C.c_void_p(), C.c_size_t(0)
    L.orc_bam_read_aux_map(file_bytes, len(file_bytes), int(exclude_standard), C.byref(blob), C.byref(n))
    data = C.string_at(blob, n.value)
    L.orc_free(blob)
    return decode_bcf_blob(data)
