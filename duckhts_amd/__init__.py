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
K_NAMES = ["sigscan", "huff_decode", "lz_resolve", "tiles", "core_unpack", "scan", "string_write", "bcf_check", "bcf_measure", "bcf_write"]


class StrCol(C.Structure):
    _fields_ = [("off", C.c_void_p), ("len", C.c_void_p), ("bytes", C.c_void_p), ("nbytes", C.c_uint64)]


class BamBatch(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("status", C.c_int32), ("seq_packed", C.c_int32),
                ("flag", C.c_void_p), ("pos", C.c_void_p), ("mapq", C.c_void_p), ("pnext", C.c_void_p),
                ("tlen", C.c_void_p), ("tid", C.c_void_p), ("mtid", C.c_void_p), ("rg_idx", C.c_void_p),
                ("rg_valid", C.c_void_p),
                ("qname", StrCol), ("cigar", StrCol), ("seq", StrCol), ("qual", StrCol), ("rg", StrCol),
                ("first_rec_uoff", C.c_uint64), ("end_uoff", C.c_uint64), ("n_tag_cols", C.c_int32), ("qual_bits", C.c_int32), ("aux_map", C.c_void_p), ("tag_cols", C.c_void_p),
                ("ov_off", C.c_void_p), ("ov_ids", C.c_void_p), ("n_ov", C.c_uint64)]


class AuxMap(C.Structure):
    _fields_ = [("valid", C.c_void_p), ("off", C.c_void_p), ("n_ent", C.c_uint64), ("key", C.c_void_p), ("kind", C.c_void_p), ("sub", C.c_void_p),
                ("pay_off", C.c_void_p), ("payload", C.c_void_p), ("payload_bytes", C.c_uint64)]


class BamHeader(C.Structure):
    _fields_ = [("n_ref", C.c_int32), ("ref_name", C.POINTER(C.c_char_p)), ("ref_len", C.POINTER(C.c_uint32)),
                ("text", C.POINTER(C.c_char)), ("l_text", C.c_uint32), ("n_rg", C.c_int32),
                ("rg_id", C.POINTER(C.c_char_p)), ("rg_sm", C.POINTER(C.c_char_p)), ("first_rec_uoff", C.c_uint64)]


class BcfColInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int32), ("is_list", C.c_int32), ("encoding", C.c_int32), ("reserved", C.c_int32)]


class BcfInfo(C.Structure):
    _fields_ = [("n_cols", C.c_int32), ("cols", C.POINTER(BcfColInfo)), ("n_contigs", C.c_int32), ("contig_name", C.POINTER(C.c_char_p)),
                ("n_dict", C.c_int32), ("dict_name", C.POINTER(C.c_char_p)), ("n_samples", C.c_int32), ("sample_name", C.POINTER(C.c_char_p)),
                ("tidy", C.c_int32), ("first_rec_uoff", C.c_uint64)]


class BcfCol(C.Structure):
    _fields_ = [("col", C.c_int32), ("reserved", C.c_int32), ("valid", C.c_void_p), ("fixed", C.c_void_p), ("off", C.c_void_p),
                ("bytes", C.c_void_p), ("nbytes", C.c_uint64), ("child_fixed", C.c_void_p), ("child_off", C.c_void_p), ("child_n", C.c_uint64), ("child_valid", C.c_void_p)]


class BcfBatch(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("status", C.c_int32), ("n_cols", C.c_int32), ("cols", C.POINTER(BcfCol)),
                ("first_rec_uoff", C.c_uint64), ("end_uoff", C.c_uint64)]


# DUCKDB_TYPE_* element codes -> the canonical type tags of the test oracle's column blob
_CANON_TYPE = {17: 1, 5: 2, 11: 3, 1: 4, 4: 5, 10: 6}
ENC_PLAIN, ENC_CONTIG, ENC_DICT, ENC_SAMPLE, ENC_FLOAT_TEXT = 0, 1, 2, 3, 4

EXPORTS = ["dhts_abi_version", "dhts_device_count", "dhts_create", "dhts_destroy", "dhts_error", "dhts_open_path",
           "dhts_open_host", "dhts_open_tiled", "dhts_resident_bytes", "dhts_bgzf_index", "dhts_bgzf_table",
           "dhts_bgzf_inflate_to_host", "dhts_bam_open", "dhts_bam_header_get", "dhts_bam_set_shard", "dhts_bam_set_block_range", "dhts_shard_cut",
           "dhts_bam_set_regions", "dhts_bam_load_index", "dhts_scan_window_stats", "dhts_bam_std_tag_count", "dhts_bam_std_tag_info", "dhts_bam_set_tag_columns", "dhts_bam_set_aux_map", "dhts_bam_set_overlap_intervals", "dhts_bam_set_overlap_bed", "dhts_bam_set_overlap_bed_path", "dhts_bam_build_index", "dhts_bam_index_bytes", "dhts_bam_rewind", "dhts_bam_next_batch", "dhts_memcpy_d2h", "dhts_sync", "dhts_kernel_time_ms",
           "dhts_kernel_time_reset", "dhts_set_timing", "dhts_bcf_open", "dhts_bcf_info_get", "dhts_bcf_set_projection", "dhts_bcf_set_block_range", "dhts_bcf_set_region", "dhts_bcf_load_index",
           "dhts_bcf_rewind", "dhts_bcf_next_batch",
           "dhts_open_path_range", "dhts_open_path_shard", "dhts_bam_set_file_shard", "dhts_bam_header_bytes", "dhts_voffset",
           "dhts_host_alloc", "dhts_host_free", "dhts_release_pools", "dhts_device_mem_info", "dhts_shard_window", "dhts_bcf_build_index", "dhts_bgzf_wrap", "dhts_bgzf_compress", "dhts_bgzip_file", "dhts_bgunzip_file", "dhts_bcf_is_text", "dhts_bam_set_seq_packed", "dhts_bcf_header_bytes", "dhts_bcf_region_segments", "dhts_set_super_blocks", "dhts_bam_build_index_csi", "dhts_tabix_build_index", "dhts_bcf_batch_host_bytes", "dhts_bcf_batch_fetch", "dhts_resident_from_cache", "dhts_bam_region_segments", "dhts_open_path_segments", "dhts_open_path_async", "dhts_stage_wait", "dhts_bgzf_index_staged", "dhts_blocks_ahead", "dhts_bam_batch_host_bytes", "dhts_bam_batch_fetch", "dhts_bam_batch_fetch_begin", "dhts_bam_batch_fetch_wait", "dhts_bcf_batch_fetch_begin", "dhts_bcf_batch_fetch_wait", "dhts_device_numa_node", "dhts_bind_thread_to_node", "dhts_bind_thread_near_device", "dhts_bam_set_qual_packed"]


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
        L.dhts_bam_set_regions.argtypes = [C.c_void_p, C.c_char_p]
        L.dhts_bam_std_tag_info.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
        L.dhts_bam_set_tag_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.dhts_bam_set_aux_map.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.dhts_bam_build_index.restype = C.c_int64
        L.dhts_bam_build_index.argtypes = [C.c_void_p]
        L.dhts_bam_index_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.dhts_bam_set_overlap_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.dhts_bam_load_index.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.dhts_bam_next_batch.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.POINTER(BamBatch)]
        L.dhts_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.dhts_sync.argtypes = [C.c_void_p]
        L.dhts_kernel_time_ms.restype = C.c_double
        L.dhts_kernel_time_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.dhts_kernel_time_reset.argtypes = [C.c_void_p]
        L.dhts_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.dhts_bcf_open.argtypes = [C.c_void_p, C.c_int]
        L.dhts_bcf_info_get.argtypes = [C.c_void_p, C.POINTER(BcfInfo)]
        L.dhts_bcf_set_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.dhts_bcf_set_block_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
        L.dhts_bcf_rewind.argtypes = [C.c_void_p]
        L.dhts_bcf_set_region.argtypes = [C.c_void_p, C.c_char_p]
        L.dhts_bcf_load_index.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.dhts_bcf_next_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(BcfBatch)]
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

    def bgzf_compress(self, raw, level=-1):
        """bgzip on the device: raw bytes -> the bytes of a BGZF file (EOF block included)"""
        buf = np.frombuffer(raw, dtype=np.uint8) if len(raw) else np.zeros(0, np.uint8)
        f = self.L.dhts_bgzf_compress
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
        bound = self._chk(f(self.h, buf.ctypes.data, buf.nbytes, level, None, 0))
        out = np.empty(bound, np.uint8)
        n = self._chk(f(self.h, buf.ctypes.data, buf.nbytes, level, out.ctypes.data, bound))
        return out[:n].tobytes()

    def bgzip_file(self, src, dst, level=-1):
        a, b = C.c_int64(0), C.c_int64(0)
        self.L.dhts_bgzip_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        self._chk(self.L.dhts_bgzip_file(self.h, os.fsencode(src), os.fsencode(dst), level, C.byref(a), C.byref(b)))
        return a.value, b.value

    def bgunzip_file(self, src, dst):
        a, b = C.c_int64(0), C.c_int64(0)
        self.L.dhts_bgunzip_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        self._chk(self.L.dhts_bgunzip_file(self.h, os.fsencode(src), os.fsencode(dst), C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_seq_packed(self, on=True):
        self.L.dhts_bam_set_seq_packed.argtypes = [C.c_void_p, C.c_int]
        self.L.dhts_bam_set_seq_packed.restype = None
        self.L.dhts_bam_set_seq_packed(self.h, int(on))

    def scan_window_stats(self):
        """(index windows, BGZF blocks) of the current scan range (the whole file without an index)"""
        w, b = C.c_int64(0), C.c_int64(0)
        self.L.dhts_scan_window_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        self._chk(self.L.dhts_scan_window_stats(self.h, C.byref(w), C.byref(b)))
        return w.value, b.value

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

    def set_tag_columns(self, ids):
        arr = np.array(list(ids), np.int32)
        self._chk(self.L.dhts_bam_set_tag_columns(self.h, arr.ctypes.data, len(arr)))
        self._tag_ids = list(ids)

    def build_index(self):
        """BAI bytes for the open BAM (one whole-file scan; hts_idx_push / hts_idx_finish restated on the host)"""
        n = self._chk(self.L.dhts_bam_build_index(self.h))
        out = np.empty(n, np.uint8)
        self._chk(self.L.dhts_bam_index_bytes(self.h, out.ctypes.data, n))
        return out.tobytes()

    def set_overlap_intervals(self, tid, beg, end):
        """Interval overlap join (cgranges cr_overlap semantics): intervals (tid, beg, end) half-open 0-based, tid = BAM header index."""
        tid = np.ascontiguousarray(tid, np.int32); beg = np.ascontiguousarray(beg, np.int64); end = np.ascontiguousarray(end, np.int64)
        assert len(tid) == len(beg) == len(end)
        self._chk(self.L.dhts_bam_set_overlap_intervals(self.h, tid.ctypes.data, beg.ctypes.data, end.ctypes.data, len(tid)))

    def set_overlap_bed(self, bed):
        """The join's intervals from a BED: bytes = the text itself, str / PathLike = a plain or bgzipped file.  Returns the number of
        read_bed rows; interval ids are those row numbers."""
        if isinstance(bed, (bytes, bytearray, memoryview)):
            buf = np.frombuffer(bytes(bed), dtype=np.uint8)
            self.L.dhts_bam_set_overlap_bed.restype = C.c_int64; self.L.dhts_bam_set_overlap_bed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
            n = self.L.dhts_bam_set_overlap_bed(self.h, buf.ctypes.data if buf.nbytes else None, buf.nbytes)
        else:
            self.L.dhts_bam_set_overlap_bed_path.restype = C.c_int64; self.L.dhts_bam_set_overlap_bed_path.argtypes = [C.c_void_p, C.c_char_p]
            n = self.L.dhts_bam_set_overlap_bed_path(self.h, os.fsencode(bed))
        self._chk(-1 if n < 0 else 0)
        return int(n)

    def overlap_lists(self, b):
        """(offsets u32[n_rows+1], ids u32[n_ov]) of one batch; ids are positions in the arrays given to set_overlap_intervals"""
        n = int(b.n_rows)
        if not b.ov_off:                               # join switched off (no intervals): every row has an empty list
            return np.zeros(n + 1, np.uint32), np.zeros(0, np.uint32)
        return self.d2h(b.ov_off, n + 1, np.uint32), self.d2h(b.ov_ids, int(b.n_ov), np.uint32)

    def set_aux_map(self, enable=True, exclude_standard=True):
        self._chk(self.L.dhts_bam_set_aux_map(self.h, int(enable), int(exclude_standard)))

    def aux_table(self, b):
        """AUXILIARY_TAGS of one batch: typed device entries -> keys / rendered value text (bam_aux_to_string, src/bam_reader.c:140-183;
        the %g / %lld text is host-side formatting) as two LIST(VARCHAR) columns in the canonical layout"""
        import struct as st_
        n = int(b.n_rows)
        am = C.cast(b.aux_map, C.POINTER(AuxMap)).contents
        ne = int(am.n_ent)
        valid = self.d2h(am.valid, n, np.uint8) if n else np.zeros(0, np.uint8)
        off = self.d2h(am.off, n + 1, np.uint32).astype(np.uint64) if n else np.zeros(1, np.uint64)
        key = self.d2h(am.key, ne, np.uint16)
        kind = self.d2h(am.kind, ne, np.uint8)
        sub = self.d2h(am.sub, ne, np.uint8)
        po = self.d2h(am.pay_off, ne + 1, np.uint32) if n else np.zeros(1, np.uint32)
        pay = self.d2h(am.payload, int(am.payload_bytes), np.uint8).tobytes()
        keys, vals = [], []
        for i in range(ne):
            kb = bytes([int(key[i]) & 0xff, int(key[i]) >> 8]).split(b"\0")[0]
            p = pay[int(po[i]):int(po[i + 1])]
            k = int(kind[i])
            if k == 0:
                v = b"%d" % st_.unpack("<q", p)[0]
            elif k == 1:
                v = (b"%g" % st_.unpack("<d", p)[0])
            elif k in (2, 3):
                v = p
            elif k == 4:
                v = bytes([int(sub[i])]) + b"".join(b",%d" % x for x in st_.unpack("<%dq" % (len(p) // 8), p))
            elif k == 5:
                v = bytes([int(sub[i])]) + b"".join(b",%g" % x for x in st_.unpack("<%dd" % (len(p) // 8), p))
            else:
                v = b""
            keys.append(kb)
            vals.append(v.split(b"\0")[0])                    # assigned through the NUL-terminated API
        cols = []
        for name, items in (("AUX_KEYS", keys), ("AUX_VALUES", vals)):
            lens = np.array([len(x) for x in items], np.int64)
            cols.append({"name": name, "type": 1, "is_list": 1, "valid": valid, "loff": off[:-1].copy(), "llen": off[1:] - off[:-1], "child_n": ne,
                         "csoff": np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64), "csbytes": np.frombuffer(b"".join(items), np.uint8)})
        return {"n_rows": n, "cols": cols}

    def tag_table(self, b):
        """standard-tag columns of one batch -> canonical column table (layout of tests/orc.py decode_bcf_blob)"""
        n = int(b.n_rows)
        cols = []
        arr = C.cast(b.tag_cols, C.POINTER(BcfCol))
        for i in range(b.n_tag_cols):
            dc = arr[i]
            name, ty, _ = std_tags()[dc.col]
            c = {"name": name, "type": 2 if ty in "iB" else 1, "is_list": 1 if ty == "B" else 0}
            c["valid"] = self.d2h(dc.valid, n, np.uint8) if n else np.zeros(0, np.uint8)
            if ty == "i":
                c["fixed"] = self.d2h(dc.fixed, n, np.uint64) if n else np.zeros(0, np.uint64)
            elif ty == "B":
                off = self.d2h(dc.off, n + 1, np.uint32).astype(np.uint64) if n else np.zeros(1, np.uint64)
                c["loff"], c["llen"], c["child_n"] = off[:-1].copy(), off[1:] - off[:-1], int(dc.child_n)
                c["cfixed"] = self.d2h(dc.child_fixed, int(dc.child_n), np.uint64)
            else:
                c["soff"] = self.d2h(dc.off, n + 1, np.uint32).astype(np.uint64) if n else np.zeros(1, np.uint64)
                c["sbytes"] = self.d2h(dc.bytes, int(dc.nbytes), np.uint8)
            cols.append(c)
        return {"n_rows": n, "status": int(b.status), "cols": cols}

    def set_regions(self, regions):
        """region := 'chr:beg-end,...'; returns False when no region names a known reference"""
        return self._chk(self.L.dhts_bam_set_regions(self.h, regions.encode() if isinstance(regions, str) else regions)) == 0

    def load_index(self, bai_bytes):
        buf = np.frombuffer(bai_bytes, dtype=np.uint8)
        self._chk(self.L.dhts_bam_load_index(self.h, buf.ctypes.data, buf.nbytes))

    def region_segments(self, index_bytes, cap=4096):
        """file byte ranges (beg[], end[]) the regions set on this context need, or None for the whole file (dhts_bam_region_segments)"""
        buf = np.frombuffer(index_bytes, dtype=np.uint8)
        beg, end, n = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64), C.c_int64(0)
        self.L.dhts_bam_region_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        self._chk(self.L.dhts_bam_region_segments(self.h, buf.ctypes.data, buf.nbytes, beg.ctypes.data, end.ctypes.data, cap, C.byref(n)))
        return None if n.value < 0 else (beg[:n.value].copy(), end[:n.value].copy())

    def bcf_region_segments(self, regions, index_bytes, cap=4096):
        """(header bytes, beg[], end[]) a read_bcf region query stages instead of the whole file, or None (dhts_bcf_region_segments)"""
        buf = np.frombuffer(index_bytes, dtype=np.uint8)
        beg, end, n = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64), C.c_int64(0)
        self.L.dhts_bcf_header_bytes.restype = C.c_uint64
        self.L.dhts_bcf_header_bytes.argtypes = [C.c_void_p]
        self.L.dhts_bcf_region_segments.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        self._chk(self.L.dhts_bcf_region_segments(self.h, regions.encode(), buf.ctypes.data, buf.nbytes, beg.ctypes.data, end.ctypes.data, cap, C.byref(n)))
        return None if n.value < 0 else (int(self.L.dhts_bcf_header_bytes(self.h)), beg[:n.value].copy(), end[:n.value].copy())

    def open_segments(self, path, header_bytes, beg, end):
        beg, end = np.ascontiguousarray(beg, np.uint64), np.ascontiguousarray(end, np.uint64)
        self.L.dhts_open_path_segments.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64]
        self._chk(self.L.dhts_open_path_segments(self.h, os.fsencode(path), int(header_bytes), beg.ctypes.data, end.ctypes.data, len(beg)))
        return self

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
        if b.seq_packed:                               # dhts_bam_set_seq_packed: 4-bit codes, len = bases (0: "*")
            off = self.d2h(b.seq.off, n + 1, np.uint32); ln = self.d2h(b.seq.len, n, np.uint32)
            data = self.d2h(b.seq.bytes, int(b.seq.nbytes), np.uint8)
            lut = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)
            both = np.empty(2 * len(data), np.uint8); both[0::2] = lut[data >> 4]; both[1::2] = lut[data & 15]
            txt = both.tobytes()
            res["SEQ"] = [b"*" if ln[i] == 0 else txt[2 * int(off[i]):2 * int(off[i]) + int(ln[i])] for i in range(n)]
        else:
            res["SEQ"] = strs(b.seq)
        res["QUAL"] = strs(b.qual)
        res["READ_GROUP_ID"] = strs(b.rg, valid)
        rgi = self.d2h(b.rg_idx, n, np.int32)
        sm = hdr["rg_sm"]
        res["SAMPLE_ID"] = [sm[k] if (valid[i] and k >= 0 and sm[k] is not None) else None for i, k in enumerate(rgi)]
        return res


def _gather_strings(ids, names):
    """ids (int array, -1 allowed only when names has a trailing default) -> (offsets u64[n+1], bytes u8) by dictionary lookup."""
    lens = np.array([len(x) for x in names], np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)])[:-1]
    flat = np.frombuffer(b"".join(names), np.uint8)
    l = lens[ids]
    off = np.concatenate([[0], np.cumsum(l)]).astype(np.uint64)
    total = int(off[-1])
    if total == 0:
        return off, np.zeros(0, np.uint8)
    idx = np.repeat(starts[ids] - off[:-1].astype(np.int64), l) + np.arange(total)
    return off, flat[idx]


class BcfScan:
    """read_bcf over one context: schema + batches as canonical column tables (same layout tests/orc.py decodes)."""

    def __init__(self, ctx, tidy=False):
        self.ctx = ctx
        ctx._chk(ctx.L.dhts_bcf_open(ctx.h, int(tidy)))
        inf = BcfInfo()
        ctx._chk(ctx.L.dhts_bcf_info_get(ctx.h, C.byref(inf)))
        self.schema = [{"name": inf.cols[i].name.decode(), "type": inf.cols[i].type, "is_list": inf.cols[i].is_list, "encoding": inf.cols[i].encoding}
                       for i in range(inf.n_cols)]
        self._names(inf)
        self.samples = [inf.sample_name[i] for i in range(inf.n_samples)]
        self.tidy = bool(inf.tidy)
        self.first_rec_uoff = inf.first_rec_uoff
        self.projection = list(range(len(self.schema)))

    def _names(self, inf=None):
        """contig / dictionary names; a scan of VCF text adds the names records use without a header definition (ask again after a batch)"""
        if inf is None:
            inf = BcfInfo()
            self.ctx._chk(self.ctx.L.dhts_bcf_info_get(self.ctx.h, C.byref(inf)))
        self.contigs = [inf.contig_name[i] if inf.contig_name[i] is not None else b"" for i in range(inf.n_contigs)]
        self.dict_names = [inf.dict_name[i] if inf.dict_name[i] is not None else b"." for i in range(inf.n_dict)] + [b"PASS"]   # id -1 -> literal PASS

    def set_projection(self, cols):
        ids = [c if isinstance(c, int) else [s["name"] for s in self.schema].index(c) for c in cols]
        arr = np.array(ids, np.int32)
        self.ctx._chk(self.ctx.L.dhts_bcf_set_projection(self.ctx.h, arr.ctypes.data, len(ids)))
        self.projection = ids

    def set_block_range(self, b0, b1, speculative):
        self.ctx._chk(self.ctx.L.dhts_bcf_set_block_range(self.ctx.h, b0, b1, int(speculative)))

    def rewind(self):
        self.ctx._chk(self.ctx.L.dhts_bcf_rewind(self.ctx.h))

    def set_region(self, region):
        """one region (the reference chains them); False when the region yields no iterator (unknown contig)"""
        return self.ctx._chk(self.ctx.L.dhts_bcf_set_region(self.ctx.h, region.encode() if region else None)) == 0

    def load_index(self, index_bytes):
        """CSI / TBI bytes: narrows the scan window of the region set last (call after set_region)"""
        buf = np.frombuffer(index_bytes, dtype=np.uint8)
        # VCF text: the region names a sequence of the tabix index, so it is resolved here; False = the index does not know it (region skipped)
        return self.ctx._chk(self.ctx.L.dhts_bcf_load_index(self.ctx.h, buf.ctypes.data, buf.nbytes)) == 0

    def next_batch(self, max_blocks=0):
        b = BcfBatch()
        self.ctx._chk(self.ctx.L.dhts_bcf_next_batch(self.ctx.h, max_blocks, C.byref(b)))
        return b

    def batch_table(self, b):
        """device columns of one batch -> canonical table (dictionary-coded columns expanded to strings)"""
        n = int(b.n_rows)
        d2h = self.ctx.d2h
        cols = []
        self._names()
        for i in range(b.n_cols):
            dc = b.cols[i]
            sc = self.schema[dc.col]
            c = {"name": sc["name"], "type": _CANON_TYPE[sc["type"]], "is_list": sc["is_list"]}
            c["valid"] = d2h(dc.valid, n, np.uint8) if n else np.zeros(0, np.uint8)
            enc = sc["encoding"]
            names = {ENC_CONTIG: self.contigs, ENC_DICT: self.dict_names, ENC_SAMPLE: self.samples}.get(enc)
            if not sc["is_list"]:
                if enc != ENC_PLAIN:
                    ids = d2h(dc.fixed, n, np.int32).astype(np.int64)
                    c["soff"], c["sbytes"] = _gather_strings(ids, names)
                elif sc["type"] == 17:
                    c["soff"] = d2h(dc.off, n + 1, np.uint32).astype(np.uint64) if n else np.zeros(1, np.uint64)
                    c["sbytes"] = d2h(dc.bytes, int(dc.nbytes), np.uint8)
                else:
                    w = {1: np.uint8, 4: np.uint32, 10: np.uint32, 5: np.uint64, 11: np.uint64}[sc["type"]]
                    c["fixed"] = d2h(dc.fixed, n, w).astype(np.uint64)
            else:
                off = d2h(dc.off, n + 1, np.uint32).astype(np.uint64) if n else np.zeros(1, np.uint64)
                c["loff"], c["llen"] = off[:-1].copy(), off[1:] - off[:-1]
                cn = int(dc.child_n)
                c["child_n"] = cn
                if dc.child_valid:
                    c["cvalid"] = d2h(dc.child_valid, cn, np.uint8)
                if enc == ENC_FLOAT_TEXT:                 # LIST(FLOAT) delivered as text: (float)strtod per element, NaN unless the whole text converts
                    co = d2h(dc.child_off, cn + 1, np.uint32) if n else np.zeros(1, np.uint32)
                    raw = d2h(dc.bytes, int(dc.nbytes), np.uint8).tobytes()
                    cv = c.get("cvalid", np.ones(cn, np.uint8))
                    vals = np.array([c_strtof(raw[int(co[k]):int(co[k + 1])]) if cv[k] else 0.0 for k in range(cn)], np.float32)
                    c["cfixed"] = vals.view(np.uint32).astype(np.uint64)
                elif enc != ENC_PLAIN:
                    ids = d2h(dc.child_fixed, cn, np.int32).astype(np.int64)
                    c["csoff"], c["csbytes"] = _gather_strings(ids, names)
                elif sc["type"] == 17:
                    c["csoff"] = d2h(dc.child_off, cn + 1, np.uint32).astype(np.uint64) if n else np.zeros(1, np.uint64)
                    c["csbytes"] = d2h(dc.bytes, int(dc.nbytes), np.uint8)
                else:
                    c["cfixed"] = d2h(dc.child_fixed, cn, np.uint32).astype(np.uint64)
            cols.append(c)
        return {"n_rows": n, "status": int(b.status), "cols": cols}


def c_strtof(tok: bytes):
    """(float)strtod(tok, &end) in the C locale: NaN unless the whole token converts (vep_parse_float, src/vep_parser.c:222-235)"""
    global _LIBC
    try:
        _LIBC
    except NameError:
        _LIBC = C.CDLL(None)
        _LIBC.strtod.restype = C.c_double
        _LIBC.strtod.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
    if not tok or b"\0" in tok:
        return float("nan")
    buf = C.create_string_buffer(tok)
    end = C.c_char_p()
    v = _LIBC.strtod(buf, C.byref(end))
    consumed = C.cast(end, C.c_void_p).value - C.addressof(buf)
    if consumed != len(tok):
        return float("nan")
    with np.errstate(over="ignore"):
        return float(np.float32(v))


def _concat_tables(parts, schema_cols):
    if not parts:
        return None
    out = []
    for k in range(len(parts[0]["cols"])):
        cs = [p["cols"][k] for p in parts]
        c = {"name": cs[0]["name"], "type": cs[0]["type"], "is_list": cs[0]["is_list"], "valid": np.concatenate([x["valid"] for x in cs])}

        def cat_off(key, bkey):
            base, offs = 0, []
            for x in cs:
                offs.append(x[key][:-1] + np.uint64(base))
                base += int(x[key][-1])
            return np.concatenate(offs + [np.array([base], np.uint64)]), np.concatenate([x[bkey] for x in cs])

        if not c["is_list"]:
            if "fixed" in cs[0]:
                c["fixed"] = np.concatenate([x["fixed"] for x in cs])
            else:
                c["soff"], c["sbytes"] = cat_off("soff", "sbytes")
        else:
            base, lo = 0, []
            for x in cs:
                lo.append(x["loff"] + np.uint64(base))
                base += x["child_n"]
            c["loff"], c["llen"], c["child_n"] = np.concatenate(lo), np.concatenate([x["llen"] for x in cs]), base
            if "cvalid" in cs[0]:
                c["cvalid"] = np.concatenate([x["cvalid"] for x in cs])
            if "cfixed" in cs[0]:
                c["cfixed"] = np.concatenate([x["cfixed"] for x in cs])
            else:
                c["csoff"], c["csbytes"] = cat_off("csoff", "csbytes")
        out.append(c)
    return out


def read_bcf(src, tidy=False, columns=None, device=0, max_blocks=0, block_range=None, region=None, index=None):
    """Full sequential read_bcf scan (every record in file order); returns the canonical column table.
    columns: optional projection (names or schema ids), like DuckDB's projection pushdown."""
    ctx = Context(device)
    try:
        ctx.open(src)
        ctx.bgzf_index()
        sc = BcfScan(ctx, tidy)
        if columns is not None:
            sc.set_projection(columns)
        if block_range is not None:
            sc.set_block_range(*block_range)
        parts, status, first, end = [], 0, None, None
        # region := 'a,b': chained union of single-region scans in the given order (src/bcf_reader.c:1327-1345)
        passes = [None] if region is None else ([r for r in region.split(",") if r] or [None])     # no non-empty token = no region
        for rg in passes:
            if rg is not None and not sc.set_region(rg):
                continue
            if rg is not None and index is not None and not sc.load_index(index):
                continue
            while True:
                b = sc.next_batch(max_blocks)
                if b.n_rows:
                    parts.append(sc.batch_table(b))
                    first = b.first_rec_uoff if first is None else first
                end = b.end_uoff
                status = b.status
                if b.status != 0:
                    break
        proj = [sc.schema[i] for i in sc.projection]
        cols = _concat_tables(parts, proj)
        if cols is None:
            cols = []
            for s_ in proj:
                c = {"name": s_["name"], "type": _CANON_TYPE[s_["type"]], "is_list": s_["is_list"], "valid": np.zeros(0, np.uint8)}
                varchar = s_["type"] == 17
                if not s_["is_list"]:
                    if varchar:
                        c["soff"], c["sbytes"] = np.zeros(1, np.uint64), np.zeros(0, np.uint8)
                    else:
                        c["fixed"] = np.zeros(0, np.uint64)
                else:
                    c["loff"], c["llen"], c["child_n"] = np.zeros(0, np.uint64), np.zeros(0, np.uint64), 0
                    if varchar:
                        c["csoff"], c["csbytes"] = np.zeros(1, np.uint64), np.zeros(0, np.uint8)
                    else:
                        c["cfixed"] = np.zeros(0, np.uint64)
                cols.append(c)
        return {"n_rows": sum(p["n_rows"] for p in parts), "status": status, "cols": cols, "by_name": {c["name"]: c for c in cols},
                "schema": sc.schema, "samples": sc.samples, "first_rec_uoff": first, "end_uoff": end, "header_first_rec_uoff": sc.first_rec_uoff}
    finally:
        ctx.close()


_STD_TAGS = None


def std_tags():
    """the reference's standard-tag table (src/bam_reader.c:54-70) as exposed by the library: [(name, type, subtype)]"""
    global _STD_TAGS
    if _STD_TAGS is None:
        L = lib()
        out = []
        for i in range(L.dhts_bam_std_tag_count()):
            nm, ty, sub = C.create_string_buffer(3), C.create_string_buffer(1), C.create_string_buffer(1)
            L.dhts_bam_std_tag_info(i, nm, ty, sub)
            out.append((nm.value.decode(), ty.raw.decode(), sub.raw.decode() if sub.raw != b"\0" else ""))
        _STD_TAGS = out
    return _STD_TAGS


def read_bam(src, device=0, max_blocks=0, shard=None, region=None, index=None, std_tags_cols=None, aux_map=None, overlap=None, sparse=None, overlap_bed=None):
    """Full sequential scan (reference mode (i)): all rows in file order.
    region: the reference's region := string (rows filtered on the device); index: BAI bytes narrowing the scan window.
    sparse=(header_bytes, beg[], end[]) with a path: only the header blocks and those file ranges are staged (Context.region_segments)."""
    ctx = Context(device)
    try:
        if sparse is not None:
            ctx.open_segments(src, *sparse)
        else:
            ctx.open(src)
        ctx.bgzf_index()
        hdr = ctx.bam_open()
        if shard is not None:
            ctx.set_shard(*shard)
        if region is not None:
            if not ctx.set_regions(region):
                raise DhtsError(f"No reads found for region(s): {region}")
            if index is not None:
                ctx.load_index(index)
        if std_tags_cols is not None:
            ctx.set_tag_columns(std_tags_cols)
        if aux_map is not None:
            ctx.set_aux_map(True, bool(aux_map == "exclude_standard"))
        if overlap is not None:
            ctx.set_overlap_intervals(*overlap)          # (tid, beg, end) arrays: out["OVERLAPS"] = per-row arrays of interval ids
        n_bed = None
        if overlap_bed is not None:                      # BED text (bytes) or file (path): ids = read_bed row numbers
            n_bed = ctx.set_overlap_bed(overlap_bed); overlap = True
        parts, tparts, aparts, oparts = [], [], [], []
        status = 0
        while True:
            b = ctx.next_batch(max_blocks)
            if b.n_rows:
                parts.append(ctx.batch_to_host(b, hdr))
                if std_tags_cols is not None:
                    tparts.append(ctx.tag_table(b))
                if aux_map is not None:
                    aparts.append(ctx.aux_table(b))
                if overlap is not None:
                    oparts.append(ctx.overlap_lists(b))
            status = b.status
            if b.status != 0:
                break
        out = {"n_rows": sum(p["n_rows"] for p in parts), "status": status, "header": hdr}
        if std_tags_cols is not None:
            out["tags"] = {"n_rows": out["n_rows"], "cols": _concat_tables(tparts, None) or []}
        if aux_map is not None:
            out["aux"] = {"n_rows": out["n_rows"], "cols": _concat_tables(aparts, None) or []}
        if n_bed is not None:
            out["n_bed_rows"] = n_bed
        if overlap is not None:
            out["OVERLAPS"] = [ids[off[i]:off[i + 1]] for off, ids in oparts for i in range(len(off) - 1)]
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


def shard_window(path, rank, world, header_bytes):
    """(win_begin, win_end, own_end) of `rank` when `world` ranks share ONE file (dhts_open_path_shard's cut; host only, no device)."""
    a, b, t = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    L = lib()
    L.dhts_shard_window.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    if L.dhts_shard_window(os.fsencode(path), rank, world, int(header_bytes), C.byref(a), C.byref(b), C.byref(t)) != 0:
        raise ValueError("bad shard arguments or unreadable file")
    return a.value, b.value, t.value


def check_handoff(spans):
    """spans: per-rank (first_rec_uoff, end_uoff, n_rows) in rank order, absolute inflated-stream offsets.
    Adjacent shards must chain exactly: the record after rank r's last one is rank r+1's first one."""
    for r in range(len(spans) - 1):
        if spans[r][1] != spans[r + 1][0]:
            raise RuntimeError(f"shard hand-off broken between rank {r} and {r + 1}: {spans[r][1]} != {spans[r + 1][0]}")
    return sum(s[2] for s in spans)
