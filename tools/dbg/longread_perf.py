import sys, os, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bamwriter as bw, duckhts_amd, orc
rng = random.Random(1)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
t0 = time.time()
recs = []
for i in range(N):
    seq = "".join(rng.choice("ACGT") for _ in range(L))
    qual = "".join(chr(33 + rng.choice([2, 11, 25, 37])) for _ in range(L))
    nops = rng.randint(1, 200)
    # cigar that sums to L
    k = max(1, L // nops - 1)
    cig = f"{L}M" if nops == 1 else "".join(f"{k}M1I" for _ in range(nops - 1)) + f"{L - (nops - 1) * (k + 1)}M"
    recs.append(bw.record(qname=f"read{i}", flag=0, tid=0, pos=1000 + i * 50, mapq=60, cigar=cig, seq=seq, qual=qual, tags=[("RG", "Z", "g1")]))
data = bw.bam_bytes([("chr1", 100_000_000)], recs, text="@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:100000000\n@RG\tID:g1\tSM:s1\n", payload=65280, level=6)
print(f"built {len(data)/1e6:.1f} MB in {time.time()-t0:.1f}s")
exp = orc.bam_read(data)
ctx = duckhts_amd.Context(0); ctx.open(data); ctx.bgzf_index(); hdr = ctx.bam_open()
for rep in range(3):
    ctx.rewind(); ctx.set_timing(True); ctx.reset_times()
    t1 = time.time(); rows = 0
    while True:
        b = ctx.next_batch(0); rows += b.n_rows
        if b.status != 0: break
    ctx.L.dhts_sync(ctx.h); dt = time.time() - t1
    print(f"rep {rep}: rows {rows} (oracle {exp['n_rows']}) {dt*1e3:.1f} ms  -> {len(data)/dt/1e9:.2f} GB/s BGZF; kernels", {k: round(v[0], 2) for k, v in ctx.kernel_times().items() if v[0] > 0})
got = duckhts_amd.read_bam(data)
ok = all(list(got[k]) == list(exp[k]) for k in duckhts_amd.BAM_COLUMNS)
print("parity", ok)
