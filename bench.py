#!/usr/bin/env python3
"""bench.py -- read_bam full-scan throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over the resident input: BGZF block discovery ->
inflate (Huffman decode + LZ resolve + CRC-32) -> BAM record boundaries -> 13-column unpack, with the
compressed bytes already in HBM when the timed region starts and the columns left in HBM.

Workload at N=1: BASELINE.json configs[1], "read_bam full scan of a synthetic 10 GB BGZF BAM on 1xMI355X".
The file is generated in full: `reps` segments of `--unique-records` records each, record i of the whole file drawn
from (seed, i) -- every segment is different data, coordinate-sorted over the whole file.
It is written to a scratch file, staged with dhts_open_path, and removed at the end.

Besides `value` (resident input, columns left in HBM) the line carries
  * `operator`: the same file through the DuckDB table function (mini host): pread + H2D + scan + D2H + DataChunk fill,
  * `parity_sample`: CRC-32 digests of all 13 columns of one whole segment (>= 1 % of the file) from the resident scan
    against the oracle's, computed inside this run,
  * `cpu_baseline`: the oracle port timed on the host in three thread shapes (1 core; 1 scan + 2 inflate threads, the
    reference's own shape; all cores), with nproc and the CPU model.

N>1 (torchrun): weak scaling, every rank scans its own 10 GB file (own seed), no data-path collective;
`--strong`: ONE file, every rank stages and scans only its own byte window (dhts_open_path_shard).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def scratch_dir(need_bytes):
    for d in (os.environ.get("DHTS_BENCH_DIR"), "/dev/shm", tempfile.gettempdir()):
        if d and os.path.isdir(d) and shutil.disk_usage(d).free > need_bytes * 1.15 + (1 << 30):
            return d
    return tempfile.gettempdir()


def generate_file(path, synth, seed, n_u, reps, threads):
    """the whole file, segment by segment; returns (per-segment byte sizes, header bytes, inflated bytes)"""
    total_n = n_u * reps
    sizes, raw = [], 0
    buf = np.empty(n_u * 260 + (1 << 20), np.uint8)
    with open(path, "wb") as f:
        head, _ = synth.bam_segment(0, seed=seed, total_n=total_n, with_header=True, with_eof=False, threads=threads)
        f.write(head.tobytes())
        for k in range(reps):
            seg, st = synth.bam_segment(n_u, seed=seed, total_n=total_n, rec0=k * n_u, with_header=False, with_eof=False, threads=threads, out=buf)
            seg.tofile(f)
            sizes.append(int(seg.nbytes)); raw += st["raw_bytes"]
        f.write(EOF_BLOCK)
    return sizes, int(head.nbytes), raw


def crc_str(off, ln, data):
    """(crc of the u32 lengths, crc of the concatenated string bytes) of a device-layout string column"""
    ln = np.ascontiguousarray(ln, np.uint32)
    total = int(ln.sum(dtype=np.uint64))
    if total == int(off[-1]):                                  # reserved == actual everywhere: the heap is already the concatenation
        body = data[:total]
    else:
        start = off[:-1].astype(np.int64)
        cum = np.concatenate([[0], np.cumsum(ln.astype(np.int64))])[:-1]
        body = data[np.repeat(start - cum, ln) + np.arange(total, dtype=np.int64)]
    return zlib.crc32(ln.tobytes()), zlib.crc32(np.ascontiguousarray(body).tobytes())


class DigestAcc:
    """running CRC-32 digests of the 13 columns in the oracle's digest layout (oracle/dhts_oracle.c orc_bam_digest); batches may come from
    several scans (block-range shards in file order): a CRC continued over the pieces is the CRC of their concatenation"""

    def __init__(self, hdr):
        self.n = 0
        self.crc = [0] * 32
        sm = [x if x is not None else None for x in hdr["rg_sm"]]
        self.smlen = np.array([len(x) if x is not None else 0 for x in sm] + [0], np.uint32)
        self.has = np.array([1 if x is not None else 0 for x in sm] + [0], np.uint8)
        self.flat = np.frombuffer(b"".join(x or b"" for x in sm), np.uint8)
        self.starts = np.concatenate([[0], np.cumsum(self.smlen[:-1])]).astype(np.int64)
        self.nsm = len(sm)

    def _up(self, i, arr):
        self.crc[i] = zlib.crc32(np.ascontiguousarray(arr).tobytes() if not isinstance(arr, (bytes, bytearray)) else arr, self.crc[i])

    def add(self, ctx, b):
        r = int(b.n_rows)
        if not r:
            return
        self._up(2, ctx.d2h(b.flag, r, np.uint16)); self._up(3, ctx.d2h(b.tid, r, np.int32)); self._up(4, ctx.d2h(b.pos, r, np.int64))
        self._up(5, ctx.d2h(b.mapq, r, np.int32)); self._up(6, ctx.d2h(b.mtid, r, np.int32)); self._up(7, ctx.d2h(b.pnext, r, np.int64)); self._up(8, ctx.d2h(b.tlen, r, np.int64))
        rgi = ctx.d2h(b.rg_idx, r, np.int32)
        w = ctx.d2h(b.rg_valid, (r + 63) // 64, np.uint64)
        valid = np.unpackbits(w.view(np.uint8), bitorder="little")[:r].astype(np.uint8)
        strs = {}
        for k, c in (("qname", b.qname), ("cigar", b.cigar), ("seq", b.seq), ("qual", b.qual), ("rg", b.rg)):
            off = ctx.d2h(c.off, r + 1, np.uint32); ln = ctx.d2h(c.len, r, np.uint32); data = ctx.d2h(c.bytes, int(c.nbytes), np.uint8)
            total = int(ln.sum(dtype=np.uint64))
            if total != int(off[-1]):        # reserved != actual somewhere (a QUAL cut at a NUL): gather the actual bytes
                start = off[:-1].astype(np.int64); cum = np.concatenate([[0], np.cumsum(ln.astype(np.int64))])[:-1]
                data = data[np.repeat(start - cum, ln) + np.arange(total, dtype=np.int64)]
            strs[k] = (ln, data[:total])
        for i, k in enumerate(("qname", "cigar", "seq", "qual")):
            self._up(9 + 2 * i, strs[k][0]); self._up(10 + 2 * i, strs[k][1])
        self._up(17, valid); self._up(18, (strs["rg"][0] * valid).astype(np.uint32)); self._up(19, strs["rg"][1])      # (an absent RG writes no bytes on either side)
        # SAMPLE_ID: @RG index -> SM by the header dictionary (dictionary-coded column), NULL without RG / without SM
        idx = np.where(rgi >= 0, rgi, self.nsm)
        sval = valid & self.has[idx]
        sl = self.smlen[idx] * sval
        tot = int(sl.sum(dtype=np.uint64))
        cum = np.concatenate([[0], np.cumsum(sl.astype(np.int64))])[:-1]
        body = self.flat[np.repeat(self.starts[idx] - cum, sl) + np.arange(tot, dtype=np.int64)] if tot else np.zeros(0, np.uint8)
        self._up(20, sval); self._up(21, sl.astype(np.uint32)); self._up(22, body)
        self.n += r

    def result(self, status):
        out = list(self.crc)
        out[0], out[1] = self.n, status & 0xffffffff
        return out


def gpu_digest(ctx, hdr, b0, b1, speculative, max_blocks, acc=None):
    """the oracle's digest layout (oracle/dhts_oracle.c orc_bam_digest) over the rows of blocks [b0, b1) of the resident file"""
    ctx.set_block_range(b0, b1, speculative)
    acc = acc or DigestAcc(hdr)
    while True:
        b = ctx.next_batch(max_blocks)
        acc.add(ctx, b)
        if b.status != 0:
            st = 0 if b.status == 1 else int(b.status)
            break
    return acc.result(st)


def cpu_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--unique-records", type=int, default=4_000_000, help="records per generated segment")
    ap.add_argument("--target-gb", type=float, default=10.0, help="compressed size of the file per GPU (GB)")
    ap.add_argument("--max-blocks", type=int, default=24576, help="BGZF blocks per batch (24,576 is the largest that keeps in-batch offsets below 2^32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-operator", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the read_bcf (config 3) and region / overlap-join (config 5) legs behind the timed loop")
    ap.add_argument("--no-parity-sample", action="store_true")
    ap.add_argument("--strong", action="store_true", help="N>1: ONE file, every rank stages and scans its own byte window")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks with torch.distributed.run (one per GPU)")
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
    n_u = args.unique_records
    probe, _ = synth.bam_segment(200_000, seed=42, total_n=200_000, with_header=False, with_eof=False)
    reps = max(1, int(round(args.target_gb * 1e9 / (probe.nbytes * (n_u / 200_000)))))
    n_records = n_u * reps
    strong = args.strong and world > 1
    seed = 42 if strong else 42 + rank
    ncpu, cpu_model = cpu_info()
    gen_threads = max(1, min(32, ncpu // (1 if strong else min(world, max(1, ncpu)))))
    sdir = scratch_dir(args.target_gb * 1e9 * (1 if strong else 1))
    path = os.path.join(sdir, f"dhts_bench_{os.getpid() if not strong else 'strong'}_{rank if not strong else 0}.bam")
    sizes = hdr_bytes = raw_bytes = None
    if not strong or rank == 0:
        sizes, hdr_bytes, raw_bytes = generate_file(path, synth, seed, n_u, reps, gen_threads)
    if strong:
        import torch
        meta = [sizes, hdr_bytes, raw_bytes, path]
        dist.broadcast_object_list(meta, src=0)
        sizes, hdr_bytes, raw_bytes, path = meta
    gen_s = time.time() - t0
    file_bytes = hdr_bytes + sum(sizes) + len(EOF_BLOCK)

    ctx = duckhts_amd.Context(local_rank if world > 1 else 0)
    L = ctx.L
    try:
        if strong:
            import ctypes as C
            L.dhts_open_path_shard.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_uint64]
            L.dhts_bam_set_file_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.dhts_voffset.restype = C.c_uint64
            L.dhts_voffset.argtypes = [C.c_void_p, C.c_uint64]
            L.dhts_resident_bytes.restype = C.c_uint64
            ctx._chk(L.dhts_open_path_shard(ctx.h, os.fsencode(path), rank, world, hdr_bytes))
        else:
            ctx.open(path)
        staged_bytes = int(L.dhts_resident_bytes(ctx.h))
        spans = {}

        def barrier():
            L.dhts_sync(ctx.h)
            if dist is not None:
                import torch
                torch.cuda.synchronize()
                dist.barrier()

        def step():
            nb = ctx.bgzf_index()
            if strong:
                ctx._chk(L.dhts_bam_set_file_shard(ctx.h, rank, world))
            else:
                ctx.rewind()
            rows, out_bytes, first = 0, 0, None
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
        hdr = ctx.bam_open()
        for _ in range(args.warmup):
            rows, nb, out_bytes = step()
            if not strong:
                assert rows == n_records, (rows, n_records)
        ctx.set_timing(True)
        ctx.reset_times()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rows, nb, out_bytes = step()
        barrier()
        dt = time.perf_counter() - t0
        if not strong:
            assert rows == n_records, (rows, n_records)
        ktimes = ctx.kernel_times()
        ctx.set_timing(False)

        # ---- multi-rank bookkeeping ----
        if dist is not None:
            import torch
            if strong:
                # hand-off: every rank's last record ends where the next rank's first record begins (BGZF virtual offsets), and
                # the ranks' rows add up to the file's
                v = torch.tensor([int(L.dhts_voffset(ctx.h, spans.get("first", 0))) if rows else -1, int(L.dhts_voffset(ctx.h, spans.get("end", 0))) if rows else -1, rows, staged_bytes],
                                 dtype=torch.int64, device=tdev)
                allv = [torch.zeros_like(v) for _ in range(world)]
                dist.all_gather(allv, v)
                have = [a for a in allv if int(a[2]) > 0]
                assert sum(int(a[2]) for a in allv) == n_records, [a.tolist() for a in allv]
                assert all(int(have[i][1]) == int(have[i + 1][0]) for i in range(len(have) - 1)), [a.tolist() for a in allv]
                staged_total = sum(int(a[3]) for a in allv)
            else:
                sp = torch.tensor([rows, n_records], dtype=torch.int64, device=tdev)
                allsp = [torch.zeros_like(sp) for _ in range(world)]
                dist.all_gather(allsp, sp)
                assert all(int(a[0]) == int(a[1]) for a in allsp), [a.tolist() for a in allsp]
            tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
            if strong:
                total_records, total_file_bytes = float(n_records), float(file_bytes)
            else:
                tot = torch.tensor([float(n_records), float(file_bytes)], dtype=torch.float64, device=tdev)
                dist.all_reduce(tot, op=dist.ReduceOp.SUM)
                total_records, total_file_bytes = float(tot[0].item()), float(tot[1].item())
        else:
            total_records, total_file_bytes = float(n_records), float(file_bytes)

        if rank != 0:
            if dist is not None:
                dist.barrier()                      # rank 0 is still using the file (strong mode)
                dist.destroy_process_group()
            return

        sec_per_step = dt / args.steps
        value = total_records / sec_per_step
        C_ = file_bytes / n_records
        U = raw_bytes / n_records
        O = out_bytes / max(rows, 1)
        kt = {k: {"ms_total": round(v[0], 3), "launches": v[1], "ms_per_launch": round(v[0] / v[1], 4) if v[1] else None} for k, v in ktimes.items()}
        inflate_ms_step = (ktimes["huff_decode"][0] + ktimes["lz_resolve"][0]) / args.steps
        frac_of_file = 1.0 if not strong else staged_bytes / file_bytes
        bytes_per_step = (file_bytes + raw_bytes) * frac_of_file
        achieved = bytes_per_step / (inflate_ms_step * 1e-3) / 1e9 if inflate_ms_step > 0 else 0.0
        lz_n = max(ktimes["lz_resolve"][1], 1)
        # HBM traffic of the inflate stage from the committed PMC passes (rocprofv3 cannot wrap itself): bytes per BGZF block
        # measured on full-size launches of this same workload (profiles/<round>/pmc_traffic_*.json), times the blocks of one step
        traffic, traffic_src, traffic_detail, traffic_hash = None, None, None, None
        try:
            import re as _re
            cand = sorted((f for r_ in sorted(os.listdir(os.path.join(ROOT, "profiles"))) for f in
                           [os.path.join(ROOT, "profiles", r_, x) for x in os.listdir(os.path.join(ROOT, "profiles", r_)) if x.startswith("pmc_traffic_")]),
                          key=lambda f: (os.path.basename(os.path.dirname(f)), [int(t) for t in _re.findall(r"\d+", os.path.basename(f))]))   # latest round, highest version
            if cand:
                pjd = json.load(open(cand[-1]))
                # the counters belong to the kernels they were measured on: the file carries a hash of the inflate kernels' sources
                import hashlib
                hk = hashlib.sha256(b"".join(open(os.path.join(ROOT, "duckhts_amd", "csrc", f), "rb").read() for f in ("bgzf_huff_wave.hip", "bgzf_inflate.hip"))).hexdigest()[:16]
                traffic_hash = {"measured_on": pjd.get("kernel_source_sha256"), "current": hk, "match": pjd.get("kernel_source_sha256") == hk}
                pj = pjd["per_block"]
                per_blk = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in pj.values())
                per_blk_raw = sum(v["fetch_bytes_raw"] + v["write_bytes"] for v in pj.values())
                traffic = int(per_blk * nb)
                traffic_src = os.path.relpath(cand[-1], ROOT)
                traffic_detail = {"per_block_raw": round(per_blk_raw, 1), "per_block_fetch_x2": round(per_blk, 1), "algorithmic_per_block": round(bytes_per_step / nb, 1),
                                  "measured_on_blocks": int(max(v.get("blocks_sampled", 0) for v in pj.values())), "command": pjd.get("run", {}).get("cmd"),
                                  "note": "FETCH_SIZE + WRITE_SIZE of the two inflate kernels (separate rocprofv3 --pmc passes, tools/profile_round.sh); x2 = the gfx950 correction for wide reads, an upper estimate for these narrow ones; the counters also count Infinity-Cache hits"}
        except Exception:
            traffic = None
        # per-kernel entries of SURVEY 8(d): inflate (C+U)/t1, boundary (U+8)/t2 (tile scan + repair + row bases: reads the inflated stream,
        # writes an 8-byte offset per record), unpack (U+O)/t3 (fixed-width unpack, scans, string writer: reads it again, writes the columns)
        def _pk(bytes_step, ms_step):
            a_ = bytes_step / (ms_step * 1e-3) / 1e9 if ms_step > 0 else 0.0
            return {"bytes_per_step": int(bytes_step), "ms_per_step": round(ms_step, 3), "achieved": round(a_, 2), "unit": "GB/s", "frac": round(a_ / HBM_PEAK_GBPS, 5)}
        rows_f = float(rows) * frac_of_file if strong else float(rows)
        t2_ms = ktimes["tiles"][0] / args.steps
        t3_ms = (ktimes["core_unpack"][0] + ktimes["scan"][0] + ktimes["string_write"][0]) / args.steps
        per_kernel = {
            "inflate": dict(_pk(bytes_per_step, inflate_ms_step), kernels="bgzf_huff_decode_wave + bgzf_lz_resolve", bytes="C + U"),
            "huff_decode": dict(_pk(file_bytes * frac_of_file, ktimes["huff_decode"][0] / args.steps), kernels="bgzf_huff_decode_wave", bytes="C (read); tokens and literals are scratch"),
            "lz_resolve": dict(_pk(raw_bytes * frac_of_file, ktimes["lz_resolve"][0] / args.steps), kernels="bgzf_lz_resolve", bytes="U (written)"),
            "boundary": dict(_pk(raw_bytes * frac_of_file + 8.0 * rows_f, t2_ms), kernels="bam_tile_scan + bam_tile_fix + bam_tile_finalize", bytes="U + 8 per record"),
            "unpack": dict(_pk(raw_bytes * frac_of_file + float(out_bytes), t3_ms), kernels="bam_tile_unpack + scans + bam_tile_strings", bytes="U + O"),
        }
        roof = {"bound": "hbm", "kernel": "bgzf_huff_decode_wave+bgzf_lz_resolve (inflate stage)", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_unit": "bytes per step (both kernels)", "traffic_source": traffic_src, "traffic_kernel_sources": traffic_hash, "traffic_detail": traffic_detail,
                "ms_per_step": round(inflate_ms_step, 3), "bytes_per_step": int(bytes_per_step),
                "lz_resolve_ms_per_launch": round(ktimes["lz_resolve"][0] / lz_n, 4), "lz_resolve_bytes_per_launch": int(bytes_per_step * args.steps / lz_n),
                "path_frac": round(value * (C_ + 2 * U + O) / 1e9 / HBM_PEAK_GBPS / max(args.gpus, 1), 5), "per_kernel": per_kernel}

        # ---- parity over the WHOLE file + CPU baseline on one segment, inside this run ----
        # Every one of the file's `reps` segments is a BAM of its own once the header is put in front of it (records never straddle segments),
        # so the oracle digests them independently -- in child processes, a few at a time -- while this process takes the same rows out of the
        # resident file segment by segment (block range of the segment, speculative start, halo) and keeps their running CRCs: all 13 columns of
        # all records of the launch are compared with the oracle, not a sample.
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        parity, cpu = None, None
        k_s = reps // 2
        seg_off = hdr_bytes + sum(sizes[:k_s])
        if (not args.no_parity_sample or not args.no_cpu_baseline) and world == 1:
            import orc
            with open(path, "rb") as f:
                head_b = f.read(hdr_bytes)
                f.seek(seg_off)
                sample = head_b + f.read(sizes[k_s]) + EOF_BLOCK
            if not args.no_parity_sample:
                t1 = time.perf_counter()
                worker = ("import sys, json; sys.path.insert(0, %r); import orc\n"
                          "p, h, o, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])\n"
                          "f = open(p, 'rb'); d = f.read(h); f.seek(o); d += f.read(n) + bytes.fromhex(%r)\n"
                          "print(json.dumps(orc.bam_digest(d)))\n") % (os.path.join(ROOT, "tests"), EOF_BLOCK.hex())
                segs = list(range(reps)) if not os.environ.get("DHTS_BENCH_PARITY_SEGMENTS") else list(range(reps))[:int(os.environ["DHTS_BENCH_PARITY_SEGMENTS"])]
                nwork = max(1, min(8, ncpu // 8, len(segs)))
                pending, running, want = list(segs), {}, {}
                def pump(block):
                    for k in [k for k, pr in running.items() if block or pr.poll() is not None]:
                        pr = running.pop(k); out_, err_ = pr.communicate()
                        assert pr.returncode == 0, f"oracle digest of segment {k} failed: {err_[-300:]}"
                        want[k] = json.loads(out_.strip().splitlines()[-1])
                        if block:
                            break
                    while pending and len(running) < nwork:
                        k = pending.pop(0)
                        running[k] = subprocess.Popen([sys.executable, "-c", worker, path, str(hdr_bytes), str(hdr_bytes + sum(sizes[:k])), str(sizes[k])], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                pump(False)
                coff, _, _, _ = ctx.bgzf_table(int(nb))
                names = ["n_rows", "status", "FLAG", "RNAME(id)", "POS", "MAPQ", "RNEXT(id)", "PNEXT", "TLEN", "QNAME.len", "QNAME", "CIGAR.len", "CIGAR", "SEQ.len", "SEQ", "QUAL.len", "QUAL",
                         "RG.valid", "RG.len", "READ_GROUP_ID", "SM.valid", "SM.len", "SAMPLE_ID"]
                got, rows_checked = {}, 0
                for k in segs:
                    o_k = hdr_bytes + sum(sizes[:k])
                    b0 = int(np.searchsorted(coff, o_k)); b1 = int(np.searchsorted(coff, o_k + sizes[k]))
                    got[k] = gpu_digest(ctx, hdr, b0, b1, k > 0, 4096)
                    pump(False)
                while pending or running:
                    pump(True)
                for k in segs:
                    bad = [names[i] for i in range(23) if got[k][i] != want[k][i]]
                    assert not bad, f"parity: segment {k}: digests differ in {bad}: gpu {got[k][:23]} oracle {want[k][:23]}"
                    rows_checked += got[k][0]
                parity = {"rows": rows_checked, "fraction_of_file": round(rows_checked / n_records, 4), "columns": 13, "digest": "CRC-32 of each column's values (lengths + bytes for strings, validity for nullable ones)",
                          "segments": len(segs), "of_segments": reps, "equal": True, "seconds": round(time.perf_counter() - t1, 1), "oracle_processes": nwork,
                          "how": "every segment's rows from the resident 10 GB scan (the segment's block range, speculative start, halo) vs the oracle on header + that segment; oracle digests in child processes"}
            if not args.no_cpu_baseline:
                zl = orc.use_system_zlib(True)
                modes = {}
                for name, thr in (("1_core", 0), ("1_scan_plus_2_inflate_threads", 2), ("all_cores", max(1, ncpu - 1))):
                    t1 = time.perf_counter()
                    got_n, stt = orc.bam_scan_count(sample) if thr == 0 else orc.bam_scan_count_mt(sample, thr)
                    cdt = time.perf_counter() - t1
                    assert got_n == n_u and stt == 0, (name, got_n, stt)
                    modes[name] = {"records_per_s": round(n_u / cdt, 1), "threads": 1 if thr == 0 else thr + 1, "seconds": round(cdt, 2)}
                orc.use_system_zlib(False)
                ref = modes["1_scan_plus_2_inflate_threads"]
                cpu = {"value": ref["records_per_s"], "unit": "records/s", "cores": 3, "kind": "port",
                       "sample": f"segment {k_s} of the same file ({n_u} records, {len(sample)} compressed bytes), all 13 columns materialised, inflate+crc32 via "
                                 f"{'system zlib (the reference dependency)' if zl else 'the RFC 1951 restatement'}; value = the reference's own thread shape "
                                 f"(1 scan thread + hts_set_threads(fp, 2), src/bam_reader.c:584,625)",
                       "modes": modes, "nproc": ncpu, "cpu_model": cpu_model}
    finally:
        ctx.close()

    # ---- the operator: the same file through the DuckDB table function (pread + H2D + scan + D2H + DataChunk fill) ----
    operator = None
    if not args.no_operator and world == 1:
        host = os.path.join(ROOT, "tests", "minihost", "minihost")
        thr = max(1, min(16, ncpu - 2))                       # (16 = one GPU's share of the box's cores; profiles/r04/fill_threads_r04.jsonl: 132 M records/s with 8 fill threads, 182 M with 16 and 32)
        qual_mode = {}
        def host_run(extra_env, repeats):
            env = dict(os.environ, DHTS_THREADS=str(thr), DHTS_TRACE="1", **extra_env)
            r = subprocess.run([host, duckhts_amd.LIB_PATH, "read_bam", path, "-t", str(thr), "-r", str(repeats)], capture_output=True, text=True, env=env)
            if r.returncode != 0:
                return None, (r.stdout + r.stderr)[-300:]
            assert int(r.stdout.split("OK rows=")[1].split()[0]) == n_records
            for l in r.stderr.splitlines():                    # how QUAL crossed PCIe (data-dependent: the batch's own alphabet)
                if "QUAL over PCIe:" in l:
                    w = l.split("QUAL over PCIe:")[1].split()
                    qual_mode.update({"batches_2bit": int(w[0]), "batches_4bit": int(w[5]), "batches_as_characters": int(w[9])})
            return [float(l.split("seconds=")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("RUN ")], None
        # (a) every query reads the file and copies it to the device (DHTS_FILE_CACHE=0); (b) default: the file staged by the first query
        # is still resident in HBM when the second one runs
        runs, err = host_run({"DHTS_FILE_CACHE": "0"}, 2)
        if runs:
            operator = {"records_per_s": round(n_records / runs[-1], 1), "bgzf_GBps": round(file_bytes / runs[-1] / 1e9, 3), "seconds": round(runs[-1], 3),
                        "first_query_seconds": round(runs[0], 3), "includes": "pread+H2D+scan+D2H+fill", "columns": 13, "DHTS_THREADS": thr,
                        "qual_over_pcie": dict(qual_mode, note="QUAL crosses PCIe as 2- / 4-bit codes when a batch holds at most 4 / 16 distinct characters (this synthetic file: 4 quality bins, as current Illumina instruments write; a file with more distinct qualities travels as characters and gains nothing here), SEQ as the file's 4-bit codes"),
                        "how": "read_bam(path) through duckhts_init_c_api driven by the mini DuckDB host (tests/minihost), second query of one process, DHTS_FILE_CACHE=0"}
            runs2, err2 = host_run({}, 3)          # (the second query re-sizes the pinned arenas once: batches are full-size from the start when nothing is staged)
            if runs2:
                operator["file_resident"] = {"records_per_s": round(n_records / runs2[-1], 1), "seconds": round(runs2[-1], 3), "includes": "scan+D2H+fill (the file staged by the previous query is still in HBM)"}
        else:
            operator = {"error": err}

    # ---- configs 3 and 5 (BASELINE.json): read_bcf on a synthetic 1 GB 16-sample BCF; region queries and the overlap join on a BAM ----
    # Run by the tools that profiles/ quotes, as child processes after the timed loop (their own contexts and files); one JSON line per query.
    def tool_lines(script, extra, timeout):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + extra, capture_output=True, text=True, timeout=timeout)
            if r.returncode != 0:
                return {"error": (r.stdout + r.stderr)[-300:]}
            out = {}
            for l in r.stdout.splitlines():
                if l.startswith("{"):
                    d = json.loads(l)
                    out[d.get("query", str(len(out)))] = {k: d[k] for k in ("value", "unit", "rows_per_s", "ms_per_step", "bgzf_GBps", "projected_columns", "index_windows", "blocks_scanned", "index_build_s", "rows_out", "pairs_out", "pairs_per_s", "cpu_baseline", "config", "roofline", "kernels", "output_bytes_per_record") if k in d and d[k] is not None}
            return out
        except Exception as e:                                   # (a bench line without these legs is still a bench line)
            return {"error": repr(e)[:300]}
    bcf_leg = region_leg = None
    if not args.no_extra_configs and world == 1:
        bcf_leg = tool_lines("bench_bcf.py", ["--steps", "3", "--warmup", "1", "--cpu-sample-records", "20000"], 400)
        region_leg = tool_lines("bench_overlap.py", ["--steps", "3", "--warmup", "1", "--target-gb", "1.0", "--indexed-records", "4000000",
                                                     "--queries", "region,overlap,indexed1,indexed10k"], 500)

    try:
        os.unlink(path)
    except OSError:
        pass
    if dist is not None:
        dist.barrier()

    line = {
        "metric": "read_bam_records_per_sec", "value": round(value, 1), "unit": "records/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"read_bam full scan, synthetic {file_bytes / 1e9:.2f} GB BGZF BAM {'shared by all GPUs' if strong else 'per GPU'} ({n_records} distinct records generated as {reps} segments, "
                               f"150 bp paired, coordinate-sorted, zlib-6, records straddle blocks), 13 core columns, inputs resident in HBM",
                   "records_per_gpu": n_records if not strong else n_records // world, "bgzf_blocks": int(nb), "compressed_bytes_per_record": round(C_, 2),
                   "inflated_bytes_per_record": round(U, 2), "column_bytes_per_record": round(O, 2), "batch_blocks": args.max_blocks,
                   "parallelism": f"bgzf-block-range shards x{world}" + (" of one file, byte-window staging" if strong else "")},
        "bgzf_GBps": round(total_file_bytes / sec_per_step / 1e9, 3),
        "roofline": roof, "operator": operator, "parity_sample": parity, "cpu_baseline": cpu, "kernels": kt, "gen_seconds": round(gen_s, 1),
        "bcf": bcf_leg, "region": region_leg,
    }
    if strong:
        line["staged_bytes_total"] = staged_total
        line["staged_over_file"] = round(staged_total / file_bytes, 4)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
