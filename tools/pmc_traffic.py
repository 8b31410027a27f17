#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter CSVs (one pass per counter) to per-kernel HBM traffic.

usage: pmc_traffic.py <dir-with-FETCH_SIZE-pass> <dir-with-WRITE_SIZE-pass> <out.json> [key=value ...]
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KB by rocprofv3; on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM section), so fetched bytes = 2 x FETCH_SIZE x 1024 is an UPPER estimate for
narrower patterns (stated in the output)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    rows = defaultdict(lambda: [0.0, 0, 0])
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            rows[k][0] += float(r["Counter_Value"])
            did = (r.get("Dispatch_Id"), k)
            if did not in seen:
                seen.add(did)
                rows[k][1] += 1
                rows[k][2] += int(r.get("Grid_Size") or 0)
    return rows


def main():
    fdir, wdir, out = sys.argv[1:4]
    extra = dict(a.split("=", 1) for a in sys.argv[4:])
    f, w = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    res = {"note": "bytes = counter x 1024; fetch additionally x2 (gfx950 FETCH_SIZE tallies 128-B requests as 64 B for wide coalesced reads; uncalibrated for narrow ones)",
           "run": extra, "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fl, wl = f.get(k, [0, 0, 0]), w.get(k, [0, 0, 0])
        n = max(fl[1], wl[1], 1)
        res["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": fl[0] * 1024 * 2 / n, "write_bytes_per_launch": wl[0] * 1024 / n,
                             "fetch_counter_sum": fl[0], "write_counter_sum": wl[0]}
    # per BGZF block figures of the two inflate kernels (bench.py multiplies them by the blocks of a step).  Phase B runs one 64-lane
    # workgroup per block, so its grids count the blocks; phase A is persistent since round 3 (bgzf_huff_decode_wave: a workgroup
    # decodes many blocks) and covers the same blocks as phase B in the profiled command, so it is divided by phase B's block count
    # (round 1-2 layouts: bgzf_huff_decode ran one LANE per block)
    res["per_block"] = {}
    lz_f, lz_w = f.get("bgzf_lz_resolve"), w.get("bgzf_lz_resolve")
    for k, lanes_per_block in (("bgzf_huff_decode", 1), ("bgzf_huff_decode_wave", 0), ("bgzf_lz_resolve", 64)):
        fl, wl = f.get(k), w.get(k)
        if not fl or not wl or not fl[2]:
            continue
        if lanes_per_block == 0:
            if not lz_f or not lz_w or not lz_f[2]:
                continue
            blocks_f, blocks_w = lz_f[2] / 64, lz_w[2] / 64
        else:
            blocks_f, blocks_w = fl[2] / lanes_per_block, wl[2] / lanes_per_block
        res["per_block"][k] = {"launch_blocks": blocks_f / max(fl[1], 1), "fetch_bytes_raw": fl[0] * 1024 / blocks_f,
                               "fetch_bytes_corrected": fl[0] * 1024 * 2 / blocks_f, "write_bytes": wl[0] * 1024 / blocks_w,
                               "launches_sampled": fl[1], "blocks_sampled": blocks_f}
    # which kernels these counters belong to: a hash of the inflate kernels' sources (bench.py reports whether it still matches)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        res["kernel_source_sha256"] = hashlib.sha256(b"".join(open(os.path.join(root, "duckhts_amd", "csrc", f), "rb").read() for f in ("bgzf_huff_wave.hip", "bgzf_inflate.hip"))).hexdigest()[:16]
    except OSError:
        res["kernel_source_sha256"] = None
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f"{k:32s} n={v['launches']:5d} fetch/launch={v['fetch_bytes_per_launch'] / 1e6:10.2f} MB write/launch={v['write_bytes_per_launch'] / 1e6:10.2f} MB")


if __name__ == "__main__":
    main()
