#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the phase-A kernel alone (tools/dbg/time_huff.py), two passes; prints KB per BGZF block (raw counter x 1024 / blocks)
set -uo pipefail
out="${1:-gpurun_out/pmc_huff}"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
repo="${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --pmc FETCH_SIZE -d "$repo/$out/pf" -o pf --output-format csv -- python3 "$repo/tools/dbg/time_huff.py" > "$repo/$out/pf.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$repo/$out/pw" -o pw --output-format csv -- python3 "$repo/tools/dbg/time_huff.py" > "$repo/$out/pw.log" 2>&1
python3 - "$repo/$out" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]; acc = collections.Counter()
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "huff" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
NBLK = 3 * (4096 + 32768 + 98304 + 131072)
print({k: round(v * 1024 / NBLK / 1024, 1) for k, v in acc.items()}, "KB per block (raw)")
PY
