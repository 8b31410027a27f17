"""duckhts_amd -- MI355X-native read_bam scan path (host-side Python mirror over the C ABI).

The product is `libduckhts_amd.so` (hand-written HIP for gfx950 behind include/duckhts_amd.h and
the DuckDB C-API extension entry point).  This module only loads it with ctypes and mirrors the
reference's operator surface for tests and benchmarks:

    read_bam(path_or_bytes)  ->  dict of the 13 core columns of src/bam_reader.c:514-526

There is NO CPU fallback: everything raises if the library or an MI355X device is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DHTS_LIB") or os.path.join(_HERE, "libduckhts_amd.so")   # DHTS_LIB: kernel-variant experiments
_LIB = None

BAM_COLUMNS = ["QNAME", "FLAG", "RNAME", "POS", "MAPQ", "CIGAR", "RNEXT", "PNEXT", "TLEN", "SEQ", "QUAL",
               "READ_GROUP_ID", "SAMPLE_ID"]
K_NAMES = ["sigscan", "huff_decode", "lz_resolve", "tiles", "core_unpack", "scan", "string_write"]


class StrCol(C.Structure):
    _fields_ = [("off", C.c_void_p), ("len", C.c_void_p), ("bytes", C.c_void_p), ("nbytes", C.c_uint64)]


class BamBatch(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("status", C.c_int32), ("reserved", C.c_int32),
                ("flag", C.c_void_p), ("pos", C.c_void_p), ("mapq", C.c_void_p), ("pnext", C.c_void_p),
                ("tlen", C.c_void_p), ("tid", C.c_void_p), ("mtid", C.c_void_p), ("rg_idx", C.c_void_p),
                ("rg_valid", C.c_void_p),
                ("qname", StrCol), ("cigar", StrCol), ("seq", StrCol), ("qual", StrCol), ("rg", StrCol),
                ("first_rec_uoff", C.c_uint64), ("end_uoff", C.c_uint64)]


class BamHeader(C.Structure):
    _fields_ = [("n_ref", C.c_int32), ("ref_name", C.POINTER(C.c_char_p)), ("ref_len", C.POINTER(C.c_uint32)),
                ("text", C.POINTER(C.c_char)), ("l_text", C.c_uint32), ("n_rg", C.c_int32),
                ("rg_id", C.POINTER(C.c_char_p)), ("rg_sm", C.POINTER(C.c_char_p)), ("first_rec_uoff", C.c_uint64)]


EXPORTS = ["dhts_abi_version", "dhts_device_count", "dhts_create", "dhts_destroy", "dhts_error", "dhts_open_path",
           "dhts_open_host", "dhts_open_tiled", "dhts_resident_bytes", "dhts_bgzf_index", "dhts_bgzf_table",
           "dhts_bgzf_inflate_to_host", "dhts_bam_open", "dhts_bam_header_get", "dhts_bam_set_shard", "dhts_bam_set_block_range", "dhts_shard_cut",
           "dhts_bam_rewind", "dhts_bam_next_batch", "dhts_memcpy_d2h", "dhts_sync", "dhts_kernel_time_ms",
           "dhts_kernel_time_reset", "dhts_set_timing"]


def lib():
    """Loads the HIP library; raises if it has not been built (run __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc, gfx950)")
        L = C.CDLL(LIB_PATH)
        L.dhts_create.restype = C.c_void_p
        L.dhts_create.argtypes = [C.c_int]
        L.dhts_destroy.argtypes = [C.c_void_p]
        L.dhts_error.restype = C.c_char_p
        L.dhts_error.argtypes = [C.c_void_p]
        L.dhts_open_path.argtypes = [C.c_void_p, C.c_char_p]
        L.dhts_open_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.dhts_open_tiled.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p,
                                      C.c_uint64]
        L.dhts_resident_bytes.restype = C.c_uint64
        L.dhts_resident_bytes.argtypes = [C.c_void_p]
        L.dhts_bgzf_index.restype = C.c_int64
        L.dhts_bgzf_index.argtypes = [C.c_void_p]
        L.dhts_bgzf_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.dhts_bgzf_inflate_to_host.restype = C.c_int64
        L.dhts_bgzf_inflate_to_host.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_uint64, C.c_void_p]
        L.dhts_bam_open.argtypes = [C.c_void_p]
        L.dhts_bam_header_get.argtypes = [C.c_void_p, C.POINTER(BamHeader)]
        L.dhts_bam_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.dhts_bam_set_block_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
        L.dhts_shard_cut.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.dhts_bam_rewind.argtypes = [C.c_void_p]
        L.dhts_bam_next_batch.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.POINTER(BamBatch)]
        L.dhts_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.dhts_sync.argtypes = [C.c_void_p]
        L.dhts_kernel_time_ms.restype = C.c_double
        L.dhts_kernel_time_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.dhts_kernel_time_reset.argtypes = [C.c_void_p]
        L.dhts_set_timing.argtypes = [C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


class DhtsError(RuntimeError):
    pass


class Context:
    """One scan context = one GPU + one HIP stream (include/duckhts_amd.h)."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = self.L.dhts_create(device)
        if not self.h:
            raise DhtsError("dhts_create failed: no MI355X device / gfx950 code object (there is no CPU fallback)")

    def close(self):
        if self.h:
            self.L.dhts_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise DhtsError(self.L.dhts_error(self.h).decode())
        return rc

    # ---- input ----
    def open(self, src):
        if isinstance(src, (bytes, bytearray, memoryview)):
            buf = np.frombuffer(src, dtype=np.uint8)
            self._chk(self.L.dhts_open_host(self.h, buf.ctypes.data, buf.nbytes))
        elif isinstance(src, np.ndarray):
            self._chk(self.L.dhts_open_host(self.h, src.ctypes.data, src.nbytes))
        else:
            self._chk(self.L.dhts_open_path(self.h, os.fsencode(src)))
        return self

    def open_tiled(self, head, body, reps, tail):
        a = [np.ascontiguousarray(np.frombuffer(x, dtype=np.uint8)) if not isinstance(x, np.ndarray) else x
             for x in (head, body, tail)]
        self._chk(self.L.dhts_open_tiled(self.h, a[0].ctypes.data, a[0].nbytes, a[1].ctypes.data, a[1].nbytes, reps,
                                         a[2].ctypes.data, a[2].nbytes))
        return self

    # ---- BGZF ----
    def bgzf_index(self):
        return self._chk(self.L.dhts_bgzf_index(self.h))

    def bgzf_table(self, n):
        coff = np.zeros(n, np.uint64)
        clen = np.zeros(n, np.uint32)
        isize = np.zeros(n, np.uint32)
        st = self.L.dhts_bgzf_table(self.h, coff.ctypes.data, clen.ctypes.data, isize.ctypes.data, n)
        return coff, clen, isize, st

    def bgzf_inflate(self, blk0, nblk, cap):
        out = np.zeros(max(cap, 1), np.uint8)
        st = np.zeros(max(nblk, 1), np.int32)
        n = self._chk(self.L.dhts_bgzf_inflate_to_host(self.h, blk0, nblk, out.ctypes.data, cap, st.ctypes.data))
        return out[:n], st[:nblk]

    # ---- read_bam ----
    def bam_open(self):
        self._chk(self.L.dhts_bam_open(self.h))
        return self.header()

    def header(self):
        h = BamHeader()
        self._chk(self.L.dhts_bam_header_get(self.h, C.byref(h)))
        return {
            "n_ref": h.n_ref,
            "ref_names": [h.ref_name[i] for i in range(h.n_ref)],
            "ref_len": [h.ref_len[i] for i in range(h.n_ref)],
            "text": C.string_at(h.text, h.l_text) if h.l_text else b"",
            "rg_id": [h.rg_id[i] for i in range(h.n_rg)],
            "rg_sm": [h.rg_sm[i] for i in range(h.n_rg)],
            "first_rec_uoff": h.first_rec_uoff,
        }

    def set_shard(self, rank, world):
        self._chk(self.L.dhts_bam_set_shard(self.h, rank, world))

    def set_block_range(self, b0, b1, speculative):
        self._chk(self.L.dhts_bam_set_block_range(self.h, b0, b1, int(speculative)))

    def rewind(self):
        self._chk(self.L.dhts_bam_rewind(self.h))

    def next_batch(self, max_blocks=0, colmask=0x1FFF):
        b = BamBatch()
        self._chk(self.L.dhts_bam_next_batch(self.h, max_blocks, colmask, C.byref(b)))
        return b

    def d2h(self, ptr, count, dtype):
        out = np.empty(count, dtype)
        if count:
            self._chk(self.L.dhts_memcpy_d2h(self.h, out.ctypes.data, ptr, out.nbytes))
        return out

    def kernel_times(self):
        res = {}
        for i, name in enumerate(K_NAMES):
            n = C.c_int64(0)
            ms = self.L.dhts_kernel_time_ms(self.h, i, C.byref(n))
            res[name] = (ms, n.value)
        return res

    def set_timing(self, on):
        self.L.dhts_set_timing(self.h, int(on))

    def reset_times(self):
        self.L.dhts_kernel_time_reset(self.h)

    # ---- host mirror of one batch: device columns -> python values ----
    def batch_to_host(self, b, hdr):
        n = b.n_rows

        def strs(col, valid=None):
            off = self.d2h(col.off, n + 1, np.uint32)
            ln = self.d2h(col.len, n, np.uint32)
            data = self.d2h(col.bytes, int(col.nbytes), np.uint8).tobytes()
            return [None if (valid is not None and not valid[i]) else data[off[i]:off[i] + ln[i]] for i in range(n)]

        res = {"n_rows": n, "status": b.status}
        if n == 0:
            for k in BAM_COLUMNS:
                res[k] = [] if k in ("QNAME", "RNAME", "CIGAR", "RNEXT", "SEQ", "QUAL", "READ_GROUP_ID", "SAMPLE_ID") else np.zeros(0)
            res["tid"] = np.zeros(0, np.int32)
            res["mtid"] = np.zeros(0, np.int32)
            return res
        res["FLAG"] = self.d2h(b.flag, n, np.uint16)
        res["POS"] = self.d2h(b.pos, n, np.int64)
        res["MAPQ"] = self.d2h(b.mapq, n, np.int32)
        res["PNEXT"] = self.d2h(b.pnext, n, np.int64)
        res["TLEN"] = self.d2h(b.tlen, n, np.int64)
        tid = self.d2h(b.tid, n, np.int32)
        mtid = self.d2h(b.mtid, n, np.int32)
        res["tid"], res["mtid"] = tid, mtid
        names = hdr["ref_names"]
        res["RNAME"] = [names[t] if t >= 0 else b"*" for t in tid]       # sam_hdr_tid2name, '*' if tid < 0
        res["RNEXT"] = [names[t] if t >= 0 else b"*" for t in mtid]      # the NAME, never '='
        words = self.d2h(b.rg_valid, (n + 63) // 64, np.uint64)
        valid = [(int(words[i >> 6]) >> (i & 63)) & 1 for i in range(n)]
        res["QNAME"] = strs(b.qname)
        res["CIGAR"] = strs(b.cigar)
        res["SEQ"] = strs(b.seq)
        res["QUAL"] = strs(b.qual)
        res["READ_GROUP_ID"] = strs(b.rg, valid)
        rgi = self.d2h(b.rg_idx, n, np.int32)
        sm = hdr["rg_sm"]
        res["SAMPLE_ID"] = [sm[k] if (valid[i] and k >= 0 and sm[k] is not None) else None for i, k in enumerate(rgi)]
        return res


def read_bam(src, device=0, max_blocks=0, shard=None):
    """Full sequential scan (reference mode (i)): all rows in file order."""
    ctx = Context(device)
    try:
        ctx.open(src)
        ctx.bgzf_index()
        hdr = ctx.bam_open()
        if shard is not None:
            ctx.set_shard(*shard)
        parts = []
        status = 0
        while True:
            b = ctx.next_batch(max_blocks)
            if b.n_rows:
                parts.append(ctx.batch_to_host(b, hdr))
            status = b.status
            if b.status != 0:
                break
        out = {"n_rows": sum(p["n_rows"] for p in parts), "status": status, "header": hdr}
        for k in BAM_COLUMNS + ["tid", "mtid"]:
            vals = [p[k] for p in parts]
            if not vals:
                out[k] = []
            elif isinstance(vals[0], np.ndarray):
                out[k] = np.concatenate(vals)
            else:
                out[k] = [x for v in vals for x in v]
        return out
    finally:
        ctx.close()


def shard_cut(coff, comp_len, rank, world):
    """Block range [b0, b1) of `rank` (same arithmetic as dhts_bam_set_shard; host only, no device needed)."""
    coff = np.ascontiguousarray(coff, dtype=np.uint64)
    b0, b1 = C.c_int64(0), C.c_int64(0)
    if lib().dhts_shard_cut(coff.ctypes.data, len(coff), int(comp_len), rank, world, C.byref(b0), C.byref(b1)) != 0:
        raise ValueError("bad shard arguments")
    return b0.value, b1.value


def check_handoff(spans):
    """spans: per-rank (first_rec_uoff, end_uoff, n_rows) in rank order, absolute inflated-stream offsets.
    Adjacent shards must chain exactly: the record after rank r's last one is rank r+1's first one."""
    for r in range(len(spans) - 1):
        if spans[r][1] != spans[r + 1][0]:
            raise RuntimeError(f"shard hand-off broken between rank {r} and {r + 1}: {spans[r][1]} != {spans[r + 1][0]}")
    return sum(s[2] for s in spans)
