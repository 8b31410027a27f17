"""Per-phase cycle split of bgzf_huff_decode_wave (needs a -DHW_DIAG build: DHTS_LIB=build/libdiag.so python tools/dbg/hw_diag.py)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import duckhts_amd  # noqa: E402
from duckhts_amd import synth  # noqa: E402

head, _ = synth.bam_segment(0, seed=42, total_n=4_000_000, with_header=True, with_eof=False)
body, st = synth.bam_segment(4_000_000, seed=42, total_n=4_000_000, with_header=False, with_eof=False)
tail = np.frombuffer(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), dtype=np.uint8)
L = C.CDLL(duckhts_amd.LIB_PATH)
L.dhts_create.restype = C.c_void_p
L.dhts_open_tiled.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
L.dhts_bgzf_index.restype = C.c_int64
L.dhts_bgzf_index.argtypes = [C.c_void_p]
L.dhts_debug_time_huff.restype = C.c_double
L.dhts_debug_time_huff.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
h = L.dhts_create(0)
L.dhts_open_tiled(C.c_void_p(h), head.ctypes.data, head.nbytes, body.ctypes.data, body.nbytes, int(os.environ.get("REPS", "4")), tail.ctypes.data, tail.nbytes)
nb = L.dhts_bgzf_index(C.c_void_p(h))
names = ["header", "tables", "pass0", "pass1", "scans", "copy", "total", "blocks", "segments", "p1_rounds", "fallbacks"]
for nblk in (2048, 65536):
    if nblk + 1 > nb:
        continue
    ms = L.dhts_debug_time_huff(C.c_void_p(h), 1, nblk, 1)
    line = f"blocks={nblk} huff_decode_wave {ms:8.3f} ms"
    if hasattr(L, "dhts_debug_hw_diag"):
        d = (C.c_ulonglong * 16)()
        L.dhts_debug_hw_diag(C.c_void_p(h), d, 1)           # discard the warm-up launch + first timed launch
        ms = L.dhts_debug_time_huff(C.c_void_p(h), 1, nblk, 1)
        L.dhts_debug_hw_diag(C.c_void_p(h), d, 1)
        v = [int(x) for x in d]
        blocks = max(v[7], 1)
        line += " | per block (cycles): " + ", ".join(f"{n} {v[i] / blocks:.0f}" for i, n in enumerate(names[:7])) + f" | hdr: stage {v[11] / blocks:.0f} cltab {v[12] / blocks:.0f} walk {v[14] / blocks:.0f} | segments/block {v[8] / blocks:.2f} p1 rounds/block {v[9] / blocks:.2f} fallbacks {v[10]} (sampled blocks {v[7]})"
    print(line, flush=True)
