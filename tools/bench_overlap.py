#!/usr/bin/env python3
"""bench_overlap.py -- BASELINE.json configs[4]: read_bam with a region filter + interval overlap join, 1 x MI355X.

One JSON line per query shape: (a) region := 10,000 seeded regions (device predicate over a full scan: the synthetic file has
no BAI -- index writers are SURVEY 8(f) item 4), (b) overlap join of every read with 1,000,000 seeded BED-like intervals,
(c) both.  Inputs resident in HBM before the timed region; columns and pair lists stay in HBM.  Not the driver's bench.
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
    ap.add_argument("--unique-records", type=int, default=4_000_000)
    ap.add_argument("--target-gb", type=float, default=2.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--intervals", type=int, default=1_000_000)
    ap.add_argument("--regions", type=int, default=10_000)
    ap.add_argument("--indexed-records", type=int, default=8_000_000, help="records of the (single-segment, coordinate-sorted) file of the indexed queries")
    ap.add_argument("--queries", default="region,overlap,both,indexed1,unindexed1,indexed10k,unindexed10k,indexed100")
    args = ap.parse_args()
    import duckhts_amd
    from duckhts_amd import synth
    n_u = args.unique_records
    head, _ = synth.bam_segment(0, seed=42, total_n=n_u, with_header=True, with_eof=False)
    body, st = synth.bam_segment(n_u, seed=42, total_n=n_u, with_header=False, with_eof=False)
    tail = np.frombuffer(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), dtype=np.uint8)
    reps = max(1, int(round(args.target_gb * 1e9 / body.nbytes)))
    n_records = n_u * reps
    for q in args.queries.split(","):
        rng = np.random.default_rng(7)                 # every query draws the same regions / intervals
        ctx = duckhts_amd.Context(0)
        if q in ("indexed1", "unindexed1", "indexed10k", "unindexed10k", "indexed100"):        # an index needs a coordinate-sorted file: one segment, not the tiled one
            sorted_file = synth.bam_file(args.indexed_records, seed=42)
            ctx.open(sorted_file)
            n_q, bytes_q = args.indexed_records, len(sorted_file)
        else:
            ctx.open_tiled(head, body, reps, tail)
            n_q, bytes_q = n_records, head.nbytes + body.nbytes * reps
        nb = ctx.bgzf_index()
        hdr = ctx.bam_open()
        names = [x.decode() for x in hdr["ref_names"]]
        lens = np.array(hdr["ref_len"], np.int64)
        build_s = None
        if q in ("indexed10k", "unindexed10k", "indexed100"):
            # BASELINE config 5: a self-built BAI and seeded regions with log-uniform widths 100 bp .. 1 Mb
            nreg = 100 if q == "indexed100" else args.regions
            if q != "unindexed10k":
                t1 = time.perf_counter(); bai = ctx.build_index(); build_s = time.perf_counter() - t1
            t = rng.choice(len(names), nreg, p=lens / lens.sum())
            w = np.exp(rng.uniform(np.log(100), np.log(1_000_000), nreg)).astype(np.int64)
            b = (rng.random(nreg) * np.maximum(lens[t] - w, 1)).astype(np.int64) + 1
            ctx.set_regions(",".join(f"{names[a]}:{s}-{s + d}" for a, s, d in zip(t, b, w)))
            if q != "unindexed10k":
                ctx.load_index(bai)
        if q in ("indexed1", "unindexed1"):
            # one 1 Mb region; "indexed1" first builds a BAI for the file (dhts_bam_build_index) and lets it narrow the scan window
            if q == "indexed1":
                t1 = time.perf_counter(); bai = ctx.build_index(); build_s = time.perf_counter() - t1
            ctx.set_regions(f"{names[1]}:20,000,000-21,000,000")
            if q == "indexed1":
                ctx.load_index(bai)
        if q in ("region", "both"):
            t = rng.choice(len(names), args.regions, p=lens / lens.sum())          # regions / intervals fall on contigs in proportion to their length
            b = (rng.random(args.regions) * np.maximum(lens[t] - 20000, 1)).astype(np.int64) + 1
            w = rng.choice([200, 1000, 5000, 20000], args.regions)
            ctx.set_regions(",".join(f"{names[a]}:{s}-{s + d}" for a, s, d in zip(t, b, w)))
        if q in ("overlap", "both"):
            t = rng.choice(len(names), args.intervals, p=lens / lens.sum()).astype(np.int32)
            b = (rng.random(args.intervals) * np.maximum(lens[t] - 5000, 1)).astype(np.int64)
            e = b + rng.choice([50, 200, 1000, 5000], args.intervals)
            ctx.set_overlap_intervals(t, b, e)

        def step():
            if q in ("region", "overlap", "both"):       # (re-indexing the BGZF container would reset the index window)
                ctx.bgzf_index()
            ctx.rewind()
            rows = pairs = 0
            while True:
                bt = ctx.next_batch(16384)
                rows += bt.n_rows
                pairs += bt.n_ov
                if bt.status != 0:
                    if bt.status < 0:
                        raise RuntimeError(f"scan ended with status {bt.status}")
                    break
            return rows, pairs
        for _ in range(args.warmup):
            rows, pairs = step()
        ctx.L.dhts_sync(ctx.h)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rows, pairs = step()
        ctx.L.dhts_sync(ctx.h)
        dt = (time.perf_counter() - t0) / args.steps
        import ctypes as C
        nw, nbk = C.c_int64(0), C.c_int64(0)
        ctx.L.dhts_scan_window_stats(C.c_void_p(ctx.h), C.byref(nw), C.byref(nbk))
        print(json.dumps({"metric": "read_bam_records_per_sec", "query": q, "value": round(n_q / dt, 1), "unit": "records/s (records scanned)",
                          "ms_per_step": round(dt * 1e3, 2), "index_windows": nw.value, "blocks_scanned": nbk.value, "index_build_s": None if build_s is None else round(build_s, 2), "rows_out": int(rows), "pairs_out": int(pairs), "pairs_per_s": round(pairs / dt, 1),
                          "config": {"workload": f"read_bam {q}: synthetic {bytes_q:,} B BGZF BAM, {n_q} records, {nb} blocks; "
                                                 f"{args.regions if q != 'overlap' else 0} regions, {args.intervals if q != 'region' else 0} intervals",
                                     "inputs": "resident in HBM", "outputs": "13 core columns + pair lists in HBM"}}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
