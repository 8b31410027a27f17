#!/usr/bin/env python3
"""PCIe-inclusive whole-operator rate: the DuckDB table function driven by the mini host on a FILE (file -> pinned host -> HBM,
GPU scan, columns copied back and written into DataChunk vectors).  Reported in DESIGN.md next to the resident-input figure;
it is never bench.py's `value`."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckhts_amd  # noqa: E402
from duckhts_amd import synth  # noqa: E402

HOST = os.path.join(ROOT, "tests", "minihost", "minihost")


def run(fn, path, proj=None, named=(), threads=1, repeat=4, env=None):
    """one host process, `repeat` queries: returns (rows, process wall time, [seconds of each query as the host measured it: bind ..
    last chunk]).  The first query pays the HIP runtime start-up, code-object load and pinned-pool allocation of a fresh process; the
    later ones are what a long-lived engine sees per query."""
    cmd = [HOST, duckhts_amd.LIB_PATH, fn, path, "-t", str(threads), "-r", str(repeat)]
    for k, v in named:
        cmd += ["-n", f"{k}={v}"]
    if proj is not None:
        cmd += ["-p", ",".join(map(str, proj))]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout + r.stderr
    if os.environ.get("DHTS_TRACE"):
        sys.stderr.write(r.stderr)
    rows = int(r.stdout.split("OK rows=")[1].split()[0])
    runs = [float(l.split("seconds=")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("RUN ")]
    return rows, dt, runs


def main_threads():
    """fill-thread scaling: rows/s of the operator against DHTS_THREADS = host threads that copy batches back and fill
    DataChunks, on a file that stays resident in HBM (DHTS_FILE_CACHE=1, warm queries) with count(*) and a fixed-width projection, so that
    neither the file read nor PCIe H2D is the limit"""
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16_000_000
    d = tempfile.mkdtemp(dir="/tmp")
    bam = os.path.join(d, "s.bam")
    synth.bam_segment(n, seed=42)[0].tofile(bam)
    size = os.path.getsize(bam)
    ncpu = os.cpu_count() or 1
    for name, proj in (("count(*) (QNAME)", [0]), ("fixed-width (FLAG,POS,MAPQ)", [1, 3, 4]), ("all 13 columns", None)):
        for thr in (1, 2, 4, 8, 16, 32):
            rows, dt, runs = run("read_bam", bam, proj=proj, threads=thr, repeat=5, env={"DHTS_THREADS": str(thr), "DHTS_FILE_CACHE": "1"})
            warm = sorted(runs[1:])[len(runs[1:]) // 2]
            print(json.dumps({"operator": "read_bam through the DuckDB table function (mini host), file resident in HBM", "projection": name, "rows": rows, "DHTS_THREADS": thr,
                              "host_cpus": ncpu, "file_GB": round(size / 1e9, 3), "warm_query_s": round(warm, 4), "records_per_s": round(rows / warm, 1)}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "threads":
        return main_threads()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    threads = int(os.environ.get("BENCH_THREADS", "8"))
    d = tempfile.mkdtemp(dir="/tmp")
    bam = os.path.join(d, "s.bam")
    synth.bam_segment(n, seed=42)[0].tofile(bam)
    size = os.path.getsize(bam)
    run("read_bam", bam, proj=[1], repeat=1)                         # warm the page cache
    for name, proj in (("count(*) (QNAME)", [0]), ("fixed-width (FLAG,POS,MAPQ)", [1, 3, 4]), ("all 13 columns", None)):
        for thr in (1, threads):
            # DHTS_FILE_CACHE=0: every query reads the file and copies it to the device; default: a file staged whole stays in HBM for the next query
            for cache in ("0", "1"):
                rows, dt, runs = run("read_bam", bam, proj=proj, threads=thr, env={"DHTS_THREADS": str(thr), "DHTS_FILE_CACHE": cache})
                warm = sorted(runs[1:])[len(runs[1:]) // 2]
                print(json.dumps({"operator": "read_bam through the DuckDB table function (mini host)",
                                  "includes": "pread + H2D + scan + D2H + chunk fill" if cache == "0" else "scan + D2H + chunk fill (file still resident in HBM from the previous query)",
                                  "projection": name, "rows": rows, "DHTS_THREADS": thr, "file_GB": round(size / 1e9, 3), "first_query_s": round(runs[0], 3), "warm_query_s": round(warm, 4),
                                  "records_per_s": round(rows / warm, 1), "bgzf_GBps": round(size / warm / 1e9, 3), "first_query_records_per_s": round(rows / runs[0], 1),
                                  "process_wall_s": round(dt, 3)}), flush=True)
    if os.environ.get("BENCH_REGION", "1") != "0":
        # region queries (config 5 through the operator): a BAI written by dhts_bam_build_index next to the file; with DHTS_SPARSE=0 the whole
        # file is staged and only the windows are inflated, by default only the header blocks and the windows are staged at all
        ctx = duckhts_amd.Context(0)
        try:
            ctx.open(bam); ctx.bgzf_index(); hdr = ctx.bam_open()
            open(bam + ".bai", "wb").write(ctx.build_index())
        finally:
            ctx.close()
        duckhts_amd.lib().dhts_release_pools()
        # the reference's only published read_bam figures are region queries on a 330 MB exome BAM (Benchmark.md:772-776, 1037-1038): COUNT(*) of
        # a region with 240,068 rows in 0.046 s, and QNAME,RNAME,POS,MAPQ,CIGAR of the same region in 0.061 s; same shape here (~240 k rows)
        regions = [("count(*), 1 region with ~240 k rows (the shape of Benchmark.md:772)", "chr1:10,000,000-10,372,000", [0]),
                   ("QNAME,RNAME,POS,MAPQ,CIGAR, same region (Benchmark.md:773)", "chr1:10,000,000-10,372,000", [0, 2, 3, 4, 5]),
                   ("all 13 columns, 1 region of 1 Mb", "chr1:10,000,000-11,000,000", None),
                   ("all 13 columns, 100 regions of 100 kb", ",".join(f"chr{1 + k % 22}:{1_000_000 * (1 + 7 * k % 40)}-{1_000_000 * (1 + 7 * k % 40) + 100_000}" for k in range(100)), None)]
        for name, region, proj in regions:
            for sparse in ("1", "0"):
                rows, dt, runs = run("read_bam", bam, proj=proj, named=[("region", region)], threads=1, repeat=6, env={"DHTS_THREADS": "1", "DHTS_SPARSE": sparse, "DHTS_FILE_CACHE": "0"})
                warm = sorted(runs[1:])[len(runs[1:]) // 2]
                print(json.dumps({"operator": "read_bam(region := ...) through the DuckDB table function (mini host)", "query": name, "rows": rows,
                                  "staging": "header + index windows" if sparse == "1" else "whole file", "file_GB": round(size / 1e9, 3),
                                  "first_query_s": round(runs[0], 3), "warm_query_ms": round(warm * 1e3, 2), "rows_per_s": round(rows / warm, 1)}), flush=True)
    if os.environ.get("BENCH_BCF", "1") == "0":
        return
    nb = n // 8
    bcf = os.path.join(d, "s.bcf")
    synth.bcf_segment(nb, seed=43)[0].tofile(bcf)
    size = os.path.getsize(bcf)
    for name, proj in (("count(*) (CHROM)", [0]), ("core 7 columns", list(range(7))), ("all 111 columns", None)):
        for thr in (1, threads):
            for cache in ("0", "1"):
                rows, dt, runs = run("read_bcf", bcf, proj=proj, threads=thr, env={"DHTS_THREADS": str(thr), "DHTS_FILE_CACHE": cache})
                warm = sorted(runs[1:])[len(runs[1:]) // 2]
                print(json.dumps({"operator": "read_bcf through the DuckDB table function (mini host)",
                                  "includes": "pread + H2D + scan + D2H + chunk fill" if cache == "0" else "scan + D2H + chunk fill (file still resident in HBM from the previous query)",
                                  "projection": name, "rows": rows, "DHTS_THREADS": thr, "file_GB": round(size / 1e9, 3),
                                  "first_query_s": round(runs[0], 3), "warm_query_s": round(warm, 4), "records_per_s": round(rows / warm, 1), "bgzf_GBps": round(size / warm / 1e9, 3)}), flush=True)


def main_producers():
    """host-side contention of several producers in one process: DHTS_DEVICES=0,0,..,0 runs k producers -- each with its
    own context, staging threads, pinned arenas and byte window of ONE file -- against a single GPU, so what changes with k is the HOST side:
    fill threads, pinned-pool mutex, page-cache reads.  Fixed-width columns (the device is not the limit) and all 13 columns, file resident
    in the page cache, DHTS_FILE_CACHE=0 (every query stages its windows again)."""
    n = int(os.environ.get("N_RECORDS", "8000000"))
    d = tempfile.mkdtemp(prefix="dhts_prod_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    bam = os.path.join(d, "p.bam"); synth.bam_segment(n, seed=11)[0].tofile(bam)
    size = os.path.getsize(bam)
    try:
        for cols, proj in (("fixed-width", [1, 3, 4, 7, 8]), ("all 13", None)):
            for k in (1, 2, 4, 8):
                for thr in (8, 16, 32):
                    env = {"DHTS_DEVICES": ",".join(["0"] * k), "DHTS_THREADS": str(thr), "DHTS_FILE_CACHE": "0"}
                    rows, dt, runs = run("read_bam", bam, proj=proj, threads=thr, repeat=4, env=env)
                    assert rows == n, (rows, n)
                    best = min(runs[1:])
                    print(json.dumps({"producers_on_one_gpu": k, "DHTS_THREADS": thr, "columns": cols, "records_per_s": round(n / best, 1), "seconds": round(best, 4),
                                      "GBps_bgzf": round(size / best / 1e9, 2), "records": n, "file_bytes": size, "nproc": len(os.sched_getaffinity(0))}), flush=True)
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "producers":
        main_producers(); sys.exit(0)
    main()
