#!/usr/bin/env python3
"""per-launch reduction of a rocprofv3 kernel trace CSV: for every kernel the full-size launches (>= half the longest one) apart from the
small ones (header inflate, last partial batch).  usage: kernel_launches.py <kt_kernel_trace.csv> [title]"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print(sys.argv[2] if len(sys.argv) > 2 else "per-launch durations (ms) from the rocprofv3 kernel trace")
print("large launches only (>= half the longest launch of the kernel); the rest are counted on the right\n")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    big = [x for x in v if x >= 0.5 * max(v)]
    print(f"{k:32s} full-size launches {len(big):4d}: avg {sum(big) / len(big):9.4f} ms  min {min(big):9.4f}  max {max(big):9.4f}   (+{len(v) - len(big)} smaller, {sum(v) - sum(big):.3f} ms)")
