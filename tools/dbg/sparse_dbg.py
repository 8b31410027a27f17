"""debug aid: one sparse region query, batch by batch"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import duckhts_amd
from duckhts_amd import synth

data = synth.bam_file(300000, seed=31)
path = "/tmp/w.bam"
open(path, "wb").write(data)
ctx = duckhts_amd.Context(0)
ctx.open(data); ctx.bgzf_index(); ctx.bam_open()
bai = ctx.build_index()
L = duckhts_amd.lib()
L.dhts_bam_header_bytes.restype = C.c_uint64
L.dhts_bam_header_bytes.argtypes = [C.c_void_p]
hb = L.dhts_bam_header_bytes(ctx.h)
region = sys.argv[1] if len(sys.argv) > 1 else "chr1:1,000,000-1,200,000"
ctx.set_regions(region)
seg = ctx.region_segments(bai)
print("header_bytes", hb, "file", len(data), "segments", [(int(a), int(b)) for a, b in zip(*seg)])
c2 = duckhts_amd.Context(0)
c2.open_segments(path, hb, seg[0], seg[1])
print("blocks", c2.bgzf_index())
c2.bam_open()
c2.set_regions(region)
c2.load_index(bai)
nw, nb = C.c_int64(), C.c_int64()
L.dhts_scan_window_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
L.dhts_scan_window_stats(c2.h, C.byref(nw), C.byref(nb))
print("windows", nw.value, "blocks", nb.value)
while True:
    b = c2.next_batch(0)
    print("batch rows", b.n_rows, "status", b.status, "first", b.first_rec_uoff, "end", b.end_uoff, "err", L.dhts_error(c2.h))
    if b.status != 0:
        break
