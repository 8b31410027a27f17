import sys, os, ctypes as C; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, duckhts_amd
from duckhts_amd import synth
import os
arr, st = synth.bam_segment(int(os.environ.get('NREC','1000000')), seed=42)
ctx = duckhts_amd.Context(0); ctx.open(arr); nb = ctx.bgzf_index(); ctx.bam_open()
rows=0
while True:
    b = ctx.next_batch(16384); rows += b.n_rows
    if b.status != 0: break
d = (C.c_ulonglong*24)()
if not hasattr(ctx.L, 'dhts_debug_diag'):
    print(rows, nb, '(no -DDHTS_DIAG build: counters only)'); sys.exit(0)
ctx.L.dhts_debug_diag(C.c_void_p(ctx.h), d)
names=['batches','rounds','easy','hard','lit_iters','far','matches','oversized']
print(rows, nb, {n:int(v) for n,v in zip(names,d)})

t=[int(x) for x in d[8:16]]
names=['crc_tables','literals','matches','token_loop','crc','flush','total','match_prep(far)']
for n,v in zip(names,t): print(n, v/nb, 'clk/block', round(100*v/max(t[6],1),1),'%')

a=[int(x) for x in d[16:24]]
print('phaseA: waves', a[4], 'iters/wave(lane0)', a[1]/max(a[4],1), 'cycles/iter', a[0]/max(a[1],1), 'sym cycles/wave', a[0]/max(a[4],1), 'total cycles/wave', a[3]/max(a[4],1))
