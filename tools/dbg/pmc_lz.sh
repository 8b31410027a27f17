#!/bin/bash
# SQ instruction mix of bgzf_lz_resolve (phase B): two rocprofv3 --pmc passes over a short bench-like scan (run on the GPU box)
set -euo pipefail
out="${1:-gpurun_out/pmc_lz}"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
repo="${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$repo/$out/p1" -o p1 --output-format csv -- python3 "$repo/tools/dbg/diag.py" > "$repo/$out/p1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU -d "$repo/$out/p2" -o p2 --output-format csv -- python3 "$repo/tools/dbg/diag.py" > "$repo/$out/p2.log" 2>&1
python3 - "$repo/$out" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        if "lz_resolve" not in k and "huff" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    w = c.get("SQ_WAVES", 1) or 1
    print(k, "per wave", {x: round(v / w, 1) for x, v in sorted(c.items())})
PY
