"""Phase-A timing experiments: python tools/dbg/time_huff.py [lib.so ...] -- ms per 65,536-block launch for each library variant."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from duckhts_amd import synth  # noqa: E402

head, _ = synth.bam_segment(0, seed=42, total_n=4_000_000, with_header=True, with_eof=False)
body, st = synth.bam_segment(4_000_000, seed=42, total_n=4_000_000, with_header=False, with_eof=False)
tail = np.frombuffer(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), dtype=np.uint8)
for lib in sys.argv[1:] or [os.path.join(ROOT, "duckhts_amd", "libduckhts_amd.so")]:
    L = C.CDLL(lib)
    L.dhts_create.restype = C.c_void_p
    L.dhts_open_tiled.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
    L.dhts_bgzf_index.restype = C.c_int64
    L.dhts_bgzf_index.argtypes = [C.c_void_p]
    L.dhts_debug_time_huff.restype = C.c_double
    L.dhts_debug_time_huff.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
    h = L.dhts_create(0)
    L.dhts_open_tiled(C.c_void_p(h), head.ctypes.data, head.nbytes, body.ctypes.data, body.nbytes, 8, tail.ctypes.data, tail.nbytes)
    nb = L.dhts_bgzf_index(C.c_void_p(h))
    for nblk in (4096, 32768, 98304, 131072):
        if nblk + 1 > nb:
            continue
        ms = L.dhts_debug_time_huff(C.c_void_p(h), 1, nblk, 2)
        print(f"{os.path.basename(lib):40s} blocks={nb} huff_decode {ms:8.3f} ms / {nblk} blocks = {ms / nblk * 65536:7.3f} ms per 65536", flush=True)
