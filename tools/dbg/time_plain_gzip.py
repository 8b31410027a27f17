#!/usr/bin/env python3
"""plain (non-BGZF) gzip through read_bcf: seconds of open + block table (= the serial device inflate + CRC check) and of the scan.
   python tools/dbg/time_plain_gzip.py [records]"""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import duckhts_amd
import test_plain_gzip as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
if len(sys.argv) > 2 and sys.argv[2] == "clinvar":                      # ClinVar-shaped lines (tools/bench_vcf_text.py): compress about 9:1, as annotation text does
    sys.path.insert(0, os.path.join(ROOT, "tools")); sys.argv = sys.argv[:1]
    import tempfile, bench_vcf_text as B
    f = os.path.join(tempfile.mkdtemp(dir="/tmp"), "c.vcf.gz"); B.generate(f, n)
    text = gzip.open(f, "rb").read()
else:
    text = T._text(n, seed=3)
for level in (1, 6):
    z = gzip.compress(text, level)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.open(z)
        t0 = time.perf_counter(); nb = ctx.bgzf_index(); ctx.L.dhts_sync(ctx.h); t1 = time.perf_counter()
        sc = duckhts_amd.BcfScan(ctx); sc.set_projection(["CHROM"])
        rows = 0
        while True:
            b = sc.next_batch(0); rows += b.n_rows
            if b.status != 0:
                break
        ctx.L.dhts_sync(ctx.h); t2 = time.perf_counter()
        print(json.dumps({"plain gzip": "level %d" % level, "text_bytes": len(text), "gzip_bytes": len(z), "rows": rows, "status": b.status, "inflate_s": round(t1 - t0, 3),
                          "inflate_MBps_out": round(len(text) / (t1 - t0) / 1e6, 1), "scan_s": round(t2 - t1, 4)}), flush=True)
    finally:
        ctx.close()
