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


def run(fn, path, proj=None, named=()):
    cmd = [HOST, duckhts_amd.LIB_PATH, fn, path]
    for k, v in named:
        cmd += ["-n", f"{k}={v}"]
    if proj is not None:
        cmd += ["-p", ",".join(map(str, proj))]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout + r.stderr
    rows = int(r.stdout.split("rows=")[1].split()[0])
    return rows, dt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    d = tempfile.mkdtemp(dir="/tmp")
    bam = os.path.join(d, "s.bam")
    synth.bam_segment(n, seed=42)[0].tofile(bam)
    size = os.path.getsize(bam)
    run("read_bam", bam, proj=[1])                                   # warm the page cache and the driver
    for name, proj in (("count(*) (QNAME)", [0]), ("fixed-width (FLAG,POS,MAPQ)", [1, 3, 4]), ("all 13 columns", None)):
        rows, dt = run("read_bam", bam, proj=proj)
        print(json.dumps({"operator": "read_bam via duckhts_init_c_api (mini host, one process, cold GPU context each run)", "projection": name, "rows": rows,
                          "file_GB": round(size / 1e9, 3), "seconds": round(dt, 3), "records_per_s": round(rows / dt, 1), "bgzf_GBps": round(size / dt / 1e9, 3)}), flush=True)
    nb = n // 8
    bcf = os.path.join(d, "s.bcf")
    synth.bcf_segment(nb, seed=43)[0].tofile(bcf)
    size = os.path.getsize(bcf)
    for name, proj in (("count(*) (CHROM)", [0]), ("core 7 columns", list(range(7))), ("all 111 columns", None)):
        rows, dt = run("read_bcf", bcf, proj=proj)
        print(json.dumps({"operator": "read_bcf via duckhts_init_c_api (mini host)", "projection": name, "rows": rows, "file_GB": round(size / 1e9, 3),
                          "seconds": round(dt, 3), "records_per_s": round(rows / dt, 1), "bgzf_GBps": round(size / dt / 1e9, 3)}), flush=True)


if __name__ == "__main__":
    main()
