#!/bin/bash
# phase-B iteration helper (GPU box): BAM parity tests, cycle split (-DDHTS_DIAG build), short bench; PMC=1 adds the SQ counter passes
tag="${1:-x}"; out="gpurun_out/ab/$tag"; mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_gpu_bam.py -x -q -m gpu > "$out/tests.txt" 2>&1; tail -2 "$out/tests.txt"
grep -q passed "$out/tests.txt" || exit 1
grep -q failed "$out/tests.txt" && exit 1
DHTS_LIB=build/lib_bdiag.so timeout -k 10 300 python tools/dbg/diag.py > "$out/diag.txt" 2>&1; head -10 "$out/diag.txt"
if [ -n "${PMC:-}" ]; then timeout -k 10 500 bash tools/dbg/pmc_lz.sh "$out/pmc" > "$out/pmc.txt" 2>&1; tail -2 "$out/pmc.txt"; fi
python bench.py --steps 5 --warmup 2 --no-extra-configs --no-operator --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; python - "$out/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["path_frac"], d["parity_sample"]["equal"])
print({k: (v["ms_per_launch"], v["launches"]) for k, v in d["kernels"].items() if v["launches"]})
PY
