"""profile aid: read_bcf over a ClinVar-shaped vcf.gz through the C ABI (run under rocprofv3 --kernel-trace --stats)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import duckhts_amd
import bench_vcf_text as B
path = "/tmp/prof.vcf.gz"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
B.generate(path, n)
data = open(path, "rb").read()
for proj in ([0], list(range(22))):
    for rep in range(3):
        t0 = time.time()
        ctx = duckhts_amd.Context(0)
        ctx.open(data); ctx.bgzf_index()
        sc = duckhts_amd.BcfScan(ctx)
        sc.set_projection(proj)
        rows = 0
        while True:
            b = sc.next_batch(0)
            rows += b.n_rows
            if b.status != 0:
                break
        ctx.close()
        print("proj", len(proj), "rows", rows, "seconds %.4f" % (time.time() - t0), flush=True)
