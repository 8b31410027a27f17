#!/usr/bin/env python3
"""bgzip on the device: throughput and compressed size against zlib on the host (one core, the same 0xff00-byte pieces).

  python tools/bench_bgzip.py [--mb 1024] [--kind vcf|bam]

Prints one JSON line per measurement: the kernel alone (input resident in HBM, dhts_kernel timing not needed: HIP events around the launch via
the library's own timer are not exposed for this path, so the wall time of dhts_bgzf_compress minus the copies is reported as well), the
buffer-to-buffer call (H2D + kernel + pack + D2H) and the file-to-file call (read + ... + write)."""
import argparse
import json
import os
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=1024)
    ap.add_argument("--kind", default="vcf")
    args = ap.parse_args()
    import duckhts_amd
    import numpy as np
    if args.kind == "vcf":
        import random
        rnd = random.Random(1)
        lines = []
        hdr = "##fileformat=VCFv4.2\n##contig=<ID=chr1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
        pos = 0
        while sum(map(len, lines[-1:])) * len(lines) < 8 << 20:
            pos += rnd.randint(1, 100)
            lines.append("chr1\t%d\trs%d\t%s\t%s\t%d\tPASS\tAC=%d;AF=%.4f;AN=%d;DP=%d;CLNSIG=%s\n" % (
                pos, rnd.randrange(10 ** 8), rnd.choice("ACGT"), rnd.choice("ACGT"), rnd.randint(10, 99), rnd.randint(1, 200), rnd.random(), rnd.randint(2, 5000), rnd.randint(1, 400),
                rnd.choice(["Benign", "Likely_benign", "Pathogenic", "Uncertain_significance"])))
        piece = (hdr + "".join(lines)).encode()
    else:
        import gzip
        from duckhts_amd import synth
        piece = gzip.decompress(synth.bam_file(60000, seed=3))
    reps = max(1, (args.mb << 20) // len(piece))
    raw = piece * reps
    n = len(raw)
    ctx = duckhts_amd.Context(0)
    try:
        ctx.bgzf_compress(raw[:1 << 20])                                    # warm-up (module load, pools)
        t0 = time.time(); z = ctx.bgzf_compress(raw); t1 = time.time()
        t2 = time.time(); z2 = ctx.bgzf_compress(raw); t3 = time.time()
        assert z == z2
        rec = {"what": "bgzip buffer to buffer (H2D + deflate + pack + D2H)", "kind": args.kind, "bytes_in": n, "bytes_out": len(z), "ratio": round(n / len(z), 3),
               "seconds": round(t3 - t2, 4), "GB_per_s_in": round(n / (t3 - t2) / 1e9, 2), "first_call_seconds": round(t1 - t0, 4)}
        print(json.dumps(rec), flush=True)
        tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        src, dst, back = os.path.join(tmp, "in"), os.path.join(tmp, "in.gz"), os.path.join(tmp, "back")
        open(src, "wb").write(raw)
        t0 = time.time(); nin, nout = ctx.bgzip_file(src, dst); t1 = time.time()
        print(json.dumps({"what": "bgzip file to file (tmpfs)", "bytes_in": nin, "bytes_out": nout, "seconds": round(t1 - t0, 4), "GB_per_s_in": round(nin / (t1 - t0) / 1e9, 2)}), flush=True)
        t0 = time.time(); a, b = ctx.bgunzip_file(dst, back); t1 = time.time()
        ok = os.path.getsize(back) == n
        print(json.dumps({"what": "bgunzip file to file (tmpfs)", "bytes_in": a, "bytes_out": b, "seconds": round(t1 - t0, 4), "GB_per_s_out": round(b / (t1 - t0) / 1e9, 2), "size_ok": ok}), flush=True)
        # round trip of the whole thing, once
        import hashlib
        assert hashlib.sha1(open(back, "rb").read()).digest() == hashlib.sha1(raw).digest()
        for f in (src, dst, back):
            os.remove(f)
        os.rmdir(tmp)
    finally:
        ctx.close()
    # host baseline: zlib on a bounded sample of the same pieces, one core
    sample = raw[:min(n, 64 << 20)]
    for lvl in (1, 6):
        t0 = time.time(); out = 0
        for k in range(0, len(sample), 65280):
            out += len(zlib.compress(sample[k:k + 65280], lvl)) + 14                 # (zlib wrapper 6 bytes vs BGZF framing 26: + 20... counted as raw deflate + 26 - 6 - 6)
        dt = time.time() - t0
        print(json.dumps({"what": f"host zlib level {lvl}, one core, same pieces", "sample_bytes": len(sample), "bytes_out": out, "ratio": round(len(sample) / out, 3), "MB_per_s_in": round(len(sample) / dt / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
