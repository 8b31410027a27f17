"""time_lz.py lib.so [lib.so ...] -- ms per bgzf_lz_resolve launch (24,576 blocks of the bench data) for library variants, e.g. the knock-out
builds -DB_EXP_NOCRC / NOLIT / NOFAR / NOROUNDS / NOREPLAY / NOSTORE of bgzf_inflate.hip: what a part of the kernel costs is the time it saves."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not os.environ.get("DHTS_LIB")):
    for lib in sys.argv[1:]:
        subprocess.call([sys.executable, os.path.abspath(__file__), "-"], env=dict(os.environ, DHTS_LIB=lib))
    sys.exit(0)
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
L.dhts_debug_time_lz.restype = C.c_double
L.dhts_debug_time_lz.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
h = L.dhts_create(0)
L.dhts_open_tiled(C.c_void_p(h), head.ctypes.data, head.nbytes, body.ctypes.data, body.nbytes, 2, tail.ctypes.data, tail.nbytes)
nb = L.dhts_bgzf_index(C.c_void_p(h))
n = min(24576, nb - 2)
ms = L.dhts_debug_time_lz(C.c_void_p(h), 1, n, 5)
print(f"{os.path.basename(duckhts_amd.LIB_PATH):28s} bgzf_lz_resolve {ms:7.3f} ms per {n} blocks", flush=True)
