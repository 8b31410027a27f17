#!/usr/bin/env python3
"""bench.py -- read_bam full-scan throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over the resident input: BGZF block discovery ->
inflate (Huffman decode + LZ resolve + CRC-32) -> BAM record boundaries -> 13-column unpack, with the
compressed bytes already in HBM when the timed region starts and the columns left in HBM.

Workload at N=1: BASELINE.json configs[1], "read_bam full scan of a synthetic 10 GB BGZF BAM on
1xMI355X".  The 10 GB are `reps` back-to-back copies of one deterministic WGS-shaped segment
(SURVEY.md 8(d) explicitly allows concatenated segments); the segment is far larger than L2+MALL,
so replays do not hit cache.  --unique-records / --target-gb change the sizes; N>1 = weak scaling,
every rank scans its own 10 GB shard (distinct seed), no data-path collective.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--unique-records", type=int, default=4_000_000, help="records in the unique segment")
    ap.add_argument("--target-gb", type=float, default=10.0, help="resident compressed size per GPU (GB)")
    ap.add_argument("--max-blocks", type=int, default=24576, help="BGZF blocks per batch (24,576 is the largest that keeps in-batch offsets below 2^32)")
    ap.add_argument("--cpu-sample-records", type=int, default=4_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="use the multi-GPU shard layout even at N=1 (testing)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # rehearsal knobs for a one-GPU box: DHTS_BENCH_BACKEND=gloo DHTS_BENCH_ONE_DEVICE=1 (all ranks share device 0)
        backend = os.environ.get("DHTS_BENCH_BACKEND", "nccl")
        if os.environ.get("DHTS_BENCH_ONE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    tdev = "cuda" if os.environ.get("DHTS_BENCH_BACKEND", "nccl") == "nccl" else "cpu"

    import duckhts_amd
    from duckhts_amd import synth

    # ---- workload (untimed) ----
    t0 = time.time()
    seed = 42 + rank
    n_u = args.unique_records
    head, _ = synth.bam_segment(0, seed=seed, total_n=n_u, with_header=True, with_eof=False)
    body, st = synth.bam_segment(n_u, seed=seed, total_n=n_u, with_header=False, with_eof=False)
    tail = np.frombuffer(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), dtype=np.uint8)
    reps = max(1, int(round(args.target_gb * 1e9 / body.nbytes)))
    gen_s = time.time() - t0
    n_records = n_u * reps
    ctx = duckhts_amd.Context(local_rank if world > 1 else 0)
    sharded = world > 1 or args.force_sharded
    if not sharded:
        ctx.open_tiled(head, body, reps, tail)
        file_bytes = head.nbytes + body.nbytes * reps + tail.nbytes
    else:
        # Every rank holds ONE SHARD of a conceptual world x 10 GB file: its block range starts in the middle of the
        # record stream (a BGZF block boundary one third into the segment, where a record straddles), so the rank
        # speculates its first record and finishes its last record from halo blocks -- the 8(e) protocol, per rank.
        b16 = body.view(np.uint8)
        offs, pos = [], 0
        while pos < body.nbytes:
            offs.append(pos)
            pos += (int(b16[pos + 16]) | (int(b16[pos + 17]) << 8)) + 1
        cut = offs[len(offs) // 3]
        halo = offs[len(offs) // 3 + 8] - cut
        ctx.open_tiled(np.concatenate([head, body[cut:]]), body, reps - 1, np.concatenate([body[:cut], body[cut:cut + halo], tail]))
        file_bytes = body.nbytes * reps
    raw_bytes = st["raw_bytes"] * reps

    def barrier():
        ctx.L.dhts_sync(ctx.h)
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    def step():
        nb = ctx.bgzf_index()
        rows = 0
        if sharded:
            ctx.set_block_range(1, 1 + n_body_blocks * reps, True)     # header block excluded; speculative first record
        else:
            ctx.rewind()
        out_bytes = 0
        first = None
        while True:
            b = ctx.next_batch(args.max_blocks)
            if b.n_rows and first is None:
                first = b.first_rec_uoff
            if b.n_rows:
                spans["first"], spans["end"] = first, b.end_uoff
            rows += b.n_rows
            out_bytes += b.n_rows * (2 + 8 + 4 + 8 + 8 + 4 + 4 + 4 + 5 * 8) + b.qname.nbytes + b.cigar.nbytes + b.seq.nbytes + b.qual.nbytes + b.rg.nbytes
            if b.status != 0:
                if b.status < 0:
                    raise RuntimeError(f"scan ended with error status {b.status}")
                break
        return rows, nb, out_bytes

    nb = ctx.bgzf_index()
    ctx.bam_open()
    spans = {}
    if sharded:
        n_body_blocks = len(offs)
    for _ in range(args.warmup):
        rows, nb, out_bytes = step()
        assert rows == n_records, (rows, n_records)
    ctx.set_timing(True)
    ctx.reset_times()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows, nb, out_bytes = step()
    barrier()
    dt = time.perf_counter() - t0
    assert rows == n_records, (rows, n_records)
    ktimes = ctx.kernel_times()
    ctx.set_timing(False)

    if sharded:
        # hand-off proof: the shard covers exactly `reps` segments of the record stream, starting and ending mid-block
        assert spans["end"] - spans["first"] == raw_bytes - 0, (spans, raw_bytes)
    if dist is not None:
        import torch
        # every rank's shard must span exactly its own `reps` segments and yield its own record count (seeds differ per rank)
        sp = torch.tensor([spans["end"] - spans["first"], raw_bytes, rows, n_records], dtype=torch.int64, device=tdev)
        allsp = [torch.zeros_like(sp) for _ in range(world)]
        dist.all_gather(allsp, sp)
        assert all(int(a[0]) == int(a[1]) and int(a[2]) == int(a[3]) for a in allsp), [a.tolist() for a in allsp]
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([float(n_records), float(file_bytes)], dtype=torch.float64, device=tdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_records, total_file_bytes = float(tot[0].item()), float(tot[1].item())
    else:
        total_records, total_file_bytes = float(n_records), float(file_bytes)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    sec_per_step = dt / args.steps
    value = total_records / sec_per_step
    C = file_bytes / n_records
    U = raw_bytes / n_records
    O = out_bytes / n_records
    # dominant kernel pair = the inflate stage; algorithmic bytes per launch = (C + U) x records per launch
    kt = {k: {"ms_total": round(v[0], 3), "launches": v[1], "ms_per_launch": round(v[0] / v[1], 4) if v[1] else None} for k, v in ktimes.items()}
    inflate_ms_step = (ktimes["huff_decode"][0] + ktimes["lz_resolve"][0]) / args.steps
    bytes_per_step = file_bytes + raw_bytes
    achieved = bytes_per_step / (inflate_ms_step * 1e-3) / 1e9 if inflate_ms_step > 0 else 0.0
    lz_n = max(ktimes["lz_resolve"][1], 1)
    # HBM traffic of the inflate stage from the committed PMC passes (rocprofv3 cannot wrap itself): bytes per BGZF block
    # measured on full-size launches of this same workload (profiles/<round>/pmc_traffic_*.json), times the blocks of one step
    traffic, traffic_src = None, None
    try:
        import re as _re
        cand = sorted((f for r_ in sorted(os.listdir(os.path.join(ROOT, "profiles"))) for f in
                       [os.path.join(ROOT, "profiles", r_, x) for x in os.listdir(os.path.join(ROOT, "profiles", r_)) if x.startswith("pmc_traffic_")]),
                      key=lambda f: (os.path.basename(os.path.dirname(f)), [int(t) for t in _re.findall(r"\d+", os.path.basename(f))]))   # latest round, highest version
        if cand:
            pj = json.load(open(cand[-1]))["per_block"]
            per_blk = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in pj.values())
            traffic = int(per_blk * nb)
            traffic_src = os.path.relpath(cand[-1], ROOT)
    except Exception:
        traffic = None
    roof = {"bound": "hbm", "kernel": "bgzf_huff_decode+bgzf_lz_resolve (inflate stage)", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_unit": "bytes per step (both kernels)", "traffic_source": traffic_src,
            "ms_per_step": round(inflate_ms_step, 3), "bytes_per_step": int(bytes_per_step),
            "lz_resolve_ms_per_launch": round(ktimes["lz_resolve"][0] / lz_n, 4), "lz_resolve_bytes_per_launch": int(bytes_per_step * args.steps / lz_n),
            "path_frac": round(value * (C + 2 * U + O) / 1e9 / HBM_PEAK_GBPS / max(args.gpus, 1), 5)}

    cpu = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        n_s = min(args.cpu_sample_records, n_u)
        sample, _ = synth.bam_segment(n_s, seed=seed, total_n=n_u)
        sb = sample.tobytes()
        zl = orc.use_system_zlib(True)
        t1 = time.perf_counter()
        got, stt = orc.bam_scan_count(sb)
        cdt = time.perf_counter() - t1
        orc.use_system_zlib(False)
        assert got == n_s and stt == 0
        cpu = {"value": round(n_s / cdt, 1), "unit": "records/s", "cores": 1, "kind": "port",
               "sample": f"first {n_s} records ({len(sb)} compressed bytes) of the same synthetic BAM, all 13 columns materialised, "
                         f"inflate+crc32 via {'system zlib (the reference dependency)' if zl else 'the RFC 1951 restatement'}, {cdt:.1f} s"}

    line = {
        "metric": "read_bam_records_per_sec", "value": round(value, 1), "unit": "records/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"read_bam full scan, synthetic {file_bytes / 1e9:.2f} GB BGZF BAM per GPU ({reps} x {n_u}-record WGS-shaped segment, "
                               f"150 bp paired, zlib-6, records straddle blocks), 13 core columns, inputs resident in HBM",
                   "records_per_gpu": n_records, "bgzf_blocks": int(nb), "compressed_bytes_per_record": round(C, 2),
                   "inflated_bytes_per_record": round(U, 2), "column_bytes_per_record": round(O, 2), "batch_blocks": args.max_blocks,
                   "parallelism": f"bgzf-block-range shards x{world}"},
        "bgzf_GBps": round(total_file_bytes / sec_per_step / 1e9, 3),
        "roofline": roof, "cpu_baseline": cpu, "kernels": kt, "gen_seconds": round(gen_s, 1),
    }
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
