#!/bin/bash
# SQ instruction mix of EVERY kernel of one bench step (run on the GPU box): one rocprofv3 --pmc pass, reduced per kernel and per wave
set -uo pipefail
out="${1:-gpurun_out/pmc_sq_bench}"; repo="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p "$repo/$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$repo/$out/p1" -o p1 --output-format csv -- python3 "$repo/bench.py" --steps 1 --warmup 0 --no-operator --no-parity-sample --no-cpu-baseline --no-extra-configs > "$repo/$out/p1.json" 2> "$repo/$out/p1.err"
python3 - "$repo/$out" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen: seen.add(key); n[k] += 1
with open(d + "/summary.txt", "w") as f:
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        w = c.get("SQ_WAVES", 1) or 1
        line = f"{k:34s} launches {n[k]:4d} waves {int(w):10d} per wave: " + " ".join(f"{x[3:]} {v / w:9.1f}" for x, v in sorted(c.items()) if x != "SQ_WAVES")
        print(line); f.write(line + "\n")
PY
find "$repo/$out" -name "*.csv" -size +4M -delete
