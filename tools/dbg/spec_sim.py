import sys, struct; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, orc
from duckhts_amd import synth
arr, st = synth.bam_segment(200000, seed=42, threads=8)
d = arr.tobytes()
z = orc.bgzf_inflate_all(d); u = z['data']; ulen=len(u)
r = orc.bam_read(d); starts = r['rec_off']; n_ref = r['n_ref']
def chk(o):
    if ulen - o < 4: return 2, 0
    bl = struct.unpack_from('<i', u, o)[0]
    if bl < 32: return 1, 0
    if ulen - o - 4 < 32: return 2, 0
    tid,pos,x2,x3,lseq,mtid,mpos,tlen = struct.unpack_from('<iiIIiiii', u, o+4)
    lq = x2 & 0xff; nc = x3 & 0xffff
    if lseq < 0 or lq < 1: return 1, 0
    if (nc<<2) + lq + ((lseq+1)>>1) + lseq > bl-32: return 1, 0
    if ulen - o - 36 < bl-32: return 2, 0
    if tid < -1 or tid >= n_ref or mtid < -1 or mtid >= n_ref: return 1, 0
    core=(nc<<2)+lq+((lseq+1)>>1)+lseq
    if (bl-32)-core > 8*core+65536: return 1,0
    if u[o+36+lq-1] != 0: return 1,0
    return 0, bl
T=8192; bad=0; tot=0
import bisect
sl = starts.tolist()
for t in range(1, ulen//T):
    tb=t*T; te=min(tb+T, ulen)
    first=None
    for o in range(tb, te):
        rc, bl = chk(o)
        if rc != 0: continue
        o2 = o+4+bl; good=True
        for k in range(2):
            rc2, bl2 = chk(o2)
            if rc2 == 1: good=False; break
            if rc2 == 2: break
            o2 += 4+bl2
        if good: first=o; break
    i = bisect.bisect_left(sl, tb)
    true = sl[i] if i < len(sl) and sl[i] < te else None
    tot+=1
    if first != true:
        bad+=1
        if bad<=10:
            print('tile',t,'spec',first,'true',true, 'delta', None if (first is None or true is None) else first-true, chk(first) if first is not None else None)
print(bad, tot)
