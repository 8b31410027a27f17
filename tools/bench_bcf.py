#!/usr/bin/env python3
"""bench_bcf.py -- read_bcf throughput on MI355X for BASELINE.json configs[2] (synthetic 1 GB 16-sample BCF).

Not the driver's bench (bench.py measures the read_bam headline metric); this prints one JSON line per query shape of
SURVEY.md 8(d) config 3: count(*) (projection {CHROM}), core-only, 6 INFO columns, all 96 FORMAT columns, everything, tidy.
Inputs are resident in HBM before the timed region; columns stay in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unique-records", type=int, default=400_000)
    ap.add_argument("--target-gb", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--max-blocks", type=int, default=24576, help="BGZF blocks per batch (profiles/r04/bcf/bcf_batch_size_*.jsonl: count(*) 17.4 / 16.8 / 16.0 ms with 4,096 / 8,192 / 24,576)")
    ap.add_argument("--queries", default="count,core,info6,format96,all,tidy")
    ap.add_argument("--cpu-sample-records", type=int, default=100_000)
    args = ap.parse_args()
    import duckhts_amd
    from duckhts_amd import synth
    n_u = args.unique_records
    head, _ = synth.bcf_segment(0, total_n=n_u, with_header=True, with_eof=False)
    body, st = synth.bcf_segment(n_u, total_n=n_u, with_header=False, with_eof=False)
    tail = np.frombuffer(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), dtype=np.uint8)
    reps = max(1, int(round(args.target_gb * 1e9 / body.nbytes)))
    n_records = n_u * reps
    file_bytes = head.nbytes + body.nbytes * reps + tail.nbytes
    raw_bytes = st["raw_bytes"] * reps
    cpu = None
    if args.cpu_sample_records:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        sb = synth.bcf_file(min(args.cpu_sample_records, n_u), total_n=n_u)
        orc.use_system_zlib(True)
        t1 = time.perf_counter(); r = orc.bcf_read(sb); cdt = time.perf_counter() - t1
        orc.use_system_zlib(False)
        cpu = {"value": round(r["n_rows"] / cdt, 1), "unit": "records/s", "cores": 1, "kind": "port",
               "sample": f"first {r['n_rows']} records, all 111 columns materialised by the oracle (system zlib inflate), {cdt:.1f} s"}
    for q in args.queries.split(","):
        ctx = duckhts_amd.Context(0)
        ctx.open_tiled(head, body, reps, tail)
        nb = ctx.bgzf_index()
        sc = duckhts_amd.BcfScan(ctx, tidy=(q == "tidy"))
        names = [s["name"] for s in sc.schema]
        proj = {"count": ["CHROM"], "core": names[:7], "info6": ["INFO_DP", "INFO_AF", "INFO_AC", "INFO_AN", "INFO_MQ", "INFO_DB"],
                "format96": [n for n in names if n.startswith("FORMAT_")], "all": names, "tidy": names}[q]
        sc.set_projection(proj)

        import ctypes as C
        ctx.L.dhts_bcf_batch_host_bytes.restype = C.c_uint64
        ctx.L.dhts_bcf_batch_host_bytes.argtypes = [C.c_void_p]
        outb = [0]

        def step():
            ctx.bgzf_index()
            sc.rewind()
            rows = 0; outb[0] = 0
            while True:
                b = sc.next_batch(args.max_blocks)
                rows += b.n_rows
                if b.n_rows:
                    outb[0] += int(ctx.L.dhts_bcf_batch_host_bytes(ctx.h))        # bytes of the batch's projected columns (validity, payloads, offsets, children)
                if b.status != 0:
                    if b.status < 0:
                        raise RuntimeError(f"scan ended with status {b.status}")
                    break
            return rows
        for _ in range(args.warmup):
            rows = step()
        ctx.set_timing(True); ctx.reset_times()
        ctx.L.dhts_sync(ctx.h)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rows = step()
        ctx.L.dhts_sync(ctx.h)
        dt = (time.perf_counter() - t0) / args.steps
        want = n_records * (16 if q == "tidy" else 1)
        assert rows == want, (rows, want)
        kt = {k: {"ms_per_step": round(v[0] / args.steps, 3), "launches_per_step": v[1] // args.steps} for k, v in ctx.kernel_times().items() if v[1]}
        # roofline of the query: B_alg = C + 2 U + O_cols -- compressed bytes read, inflated bytes written by the inflate and read by
        # the record stage, projected columns written; per-kernel entries from the HIP-event times of the same run (every kernel alone on the stream)
        PEAK = 8000.0
        kms = {k: v["ms_per_step"] for k, v in kt.items()}
        def pk(nbytes, ms):
            return {"bytes_per_step": int(nbytes), "ms_per_step": round(ms, 3), "achieved": round(nbytes / ms / 1e6, 2) if ms > 0 else None, "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / PEAK, 5) if ms > 0 else None}
        infl = kms.get("huff_decode", 0.0) + kms.get("lz_resolve", 0.0)
        rec = kms.get("tiles", 0.0) + kms.get("bcf_check", 0.0)
        cells = kms.get("bcf_measure", 0.0) + kms.get("scan", 0.0) + kms.get("bcf_write", 0.0)
        b_alg = file_bytes + 2 * raw_bytes + outb[0]
        roofline = {"bound": "hbm", "peak": PEAK, "unit": "GB/s", "bytes_per_step": int(b_alg), "bytes": "C + 2 U + O_cols", "achieved": round(b_alg / dt / 1e9, 2), "frac": round(b_alg / dt / 1e9 / PEAK, 5),
                    "device_ms_per_step": round(sum(kms.values()), 3), "wall_ms_per_step": round(dt * 1e3, 3),
                    "per_kernel": {"inflate": dict(pk(file_bytes + raw_bytes, infl), kernels="bgzf_huff_decode_wave + bgzf_lz_resolve", bytes="C + U"),
                                   "boundary+check": dict(pk(raw_bytes + 4.0 * n_records, rec), kernels="bcf_tile_scan + bcf_tile_fix + bam_tile_finalize + bcf_tile_offsets + bcf_rec_check", bytes="U + directory"),
                                   "cells": dict(pk(raw_bytes + outb[0], cells), kernels="bcf_cells<measure> + mscan_* + bcf_cells<write>", bytes="U + O_cols")}}
        print(json.dumps({"metric": "read_bcf_records_per_sec", "query": q, "value": round(n_records / dt, 1), "unit": "records/s", "rows_per_s": round(rows / dt, 1),
                          "ms_per_step": round(dt * 1e3, 2), "bgzf_GBps": round(file_bytes / dt / 1e9, 3), "projected_columns": len(proj),
                          "config": {"workload": f"read_bcf, synthetic {file_bytes / 1e9:.2f} GB BCF ({reps} x {n_u} records, 16 samples, zlib-6), inputs resident in HBM",
                                     "records": n_records, "bgzf_blocks": int(nb), "compressed_bytes_per_record": round(file_bytes / n_records, 1),
                                     "inflated_bytes_per_record": round(raw_bytes / n_records, 1), "batch_blocks": args.max_blocks},
                          "kernels": kt, "roofline": roofline, "output_bytes_per_record": round(outb[0] / n_records, 1), "cpu_baseline": cpu if q == "all" else None}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
