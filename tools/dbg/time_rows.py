"""time_rows.py -- per-launch device time of the record-stage kernels on 2 x 4 M records of the bench data (39,626 BGZF blocks, two batches),
for the three-pass row stage (default), the fused row pass (DHTS_ROWS=fused) and the timing experiments DHTS_ROWS_EXP=2 (no look-back; output invalid) and
DHTS_ROWS_WG_PER_CU (grid of the persistent kernel): what a part costs is the time it saves.  Each variant runs in a child process (the knobs are read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VARIANTS = [("three passes (default)", {}), ("fused", {"DHTS_ROWS": "fused"}), ("fused, no look-back", {"DHTS_ROWS": "fused", "DHTS_ROWS_EXP": "2"})] + [(f"fused, {k} workgroups per CU", {"DHTS_ROWS": "fused", "DHTS_ROWS_WG_PER_CU": str(k)}) for k in (4, 8, 12)]
if len(sys.argv) < 2:
    for name, env in VARIANTS:
        subprocess.call([sys.executable, os.path.abspath(__file__), name], env=dict(os.environ, **env))
    sys.exit(0)
import numpy as np  # noqa: E402
import duckhts_amd  # noqa: E402
from duckhts_amd import synth  # noqa: E402

head, _ = synth.bam_segment(0, seed=42, total_n=4_000_000, with_header=True, with_eof=False)
body, st = synth.bam_segment(4_000_000, seed=42, total_n=4_000_000, with_header=False, with_eof=False)
tail = np.frombuffer(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), dtype=np.uint8)
ctx = duckhts_amd.Context(0)
ctx.open_tiled(head, body, 2, tail)
ctx.bgzf_index(); ctx.bam_open()


def scan():
    ctx.rewind(); rows = 0
    while True:
        b = ctx.next_batch(24576)
        rows += b.n_rows
        if b.status != 0:
            return rows


scan()
ctx.set_timing(True); ctx.reset_times()
for _ in range(3):
    rows = scan()
kt = ctx.kernel_times()
print(f"{sys.argv[1]:24s} rows {rows}: " + "  ".join(f"{k} {v[0] / max(v[1], 1):.3f} ms x{v[1]}" for k, v in kt.items() if v[1] and k in ("tiles", "core_unpack", "scan", "string_write")), flush=True)
import ctypes as C  # noqa: E402
if hasattr(ctx.L, "dhts_debug_tr_diag"):
    d = (C.c_ulonglong * 16)()
    ctx.L.dhts_debug_tr_diag(C.c_void_p(ctx.h), d, 1)
    rows = scan()
    ctx.L.dhts_debug_tr_diag(C.c_void_p(ctx.h), d, 1)
    v = [int(x) for x in d]; nt = max(v[7], 1)
    names = ["arrive+commit", "record list", "phase one", "publish+ticket", "look-back", "prefetch issue", "phase two"]
    print("    cycles per tile: " + ", ".join(f"{n} {v[i] / nt:.0f}" for i, n in enumerate(names)) + f" | look-back windows per tile {v[8] / nt:.2f}, polls {v[9] / nt:.2f} (tiles {v[7]})", flush=True)
ctx.close()
