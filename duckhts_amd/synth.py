"""Deterministic synthetic BGZF BAM inputs -- workload tooling.

Thin ctypes wrapper over tools/synth_bam.c (system libz.so.1, level-6 raw deflate, 65280-byte
payloads cut without regard to record boundaries).  Not part of the scan path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "tools", "libsynth.so")
_LIB = None


def build(force=False):
    src = os.path.join(_ROOT, "tools", "synth_bam.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-pthread", "-o", _SO, src,
                               "-l:libz.so.1"])
    return _SO


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.synth_bam_segment.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.synth_bam_segment.restype = C.c_size_t
        L.synth_bcf_segment.argtypes = L.synth_bam_segment.argtypes
        L.synth_bcf_segment.restype = C.c_size_t
        _LIB = L
    return _LIB


def bam_segment(n, seed=42, total_n=None, rec0=0, with_header=True, with_eof=True, level=6, payload=65280,
                threads=None, out=None):
    """Records [rec0, rec0+n) of a conceptual total_n-record file -> (uint8 array, stats dict).

    `out` may be a preallocated uint8 numpy array (e.g. pinned memory); otherwise one is allocated.
    """
    total_n = n if total_n is None else total_n
    threads = threads or min(os.cpu_count() or 1, 32)
    cap = int(n) * 260 + (1 << 20)
    if out is None:
        out = np.empty(cap, dtype=np.uint8)
    stats = (C.c_uint64 * 2)()
    got = _lib().synth_bam_segment(seed, total_n, rec0, n, int(with_header), int(with_eof), level, payload, threads,
                                   out.ctypes.data, out.nbytes, stats)
    if got == 0:
        raise RuntimeError("synth_bam_segment: output capacity too small")
    return out[:got], {"raw_bytes": int(stats[0]), "n_blocks": int(stats[1]), "n_records": int(n)}


def bam_file(n, seed=42, **kw) -> bytes:
    arr, _ = bam_segment(n, seed=seed, **kw)
    return arr.tobytes()


def bcf_segment(n, seed=43, total_n=None, rec0=0, with_header=True, with_eof=True, level=6, payload=65280, threads=None, out=None):
    """Synthetic 16-sample BCF records [rec0, rec0+n) -> (uint8 array, stats dict)."""
    total_n = n if total_n is None else total_n
    threads = threads or min(os.cpu_count() or 1, 32)
    cap = int(n) * 700 + (1 << 20)
    if out is None:
        out = np.empty(cap, dtype=np.uint8)
    stats = (C.c_uint64 * 2)()
    got = _lib().synth_bcf_segment(seed, total_n, rec0, n, int(with_header), int(with_eof), level, payload, threads, out.ctypes.data, out.nbytes, stats)
    if got == 0:
        raise RuntimeError("synth_bcf_segment: output capacity too small")
    return out[:got], {"raw_bytes": int(stats[0]), "n_blocks": int(stats[1]), "n_records": int(n)}


def bcf_file(n, seed=43, **kw) -> bytes:
    arr, _ = bcf_segment(n, seed=seed, **kw)
    return arr.tobytes()
