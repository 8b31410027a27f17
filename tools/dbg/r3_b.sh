#!/bin/bash
# round-3 helper (GPU box), phase B: BAM parity tests, cycle split (-DDHTS_DIAG build), SQ counters, short bench
tag="${1:-b}"; out="gpurun_out/r3/$tag"; mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_gpu_bam.py -x -q -m gpu > "$out/tests.txt" 2>&1; tail -2 "$out/tests.txt"
DHTS_LIB=build/lib_bdiag.so timeout -k 10 300 python tools/dbg/diag.py > "$out/diag.txt" 2>&1; head -8 "$out/diag.txt"
timeout -k 10 500 bash tools/dbg/pmc_lz.sh "$out/pmc" > "$out/pmc.txt" 2>&1; tail -1 "$out/pmc.txt"
if [ -n "${BENCH:-}" ]; then python bench.py --steps 5 --warmup 2 --no-extra-configs --no-operator > "$out/bench.json" 2> "$out/bench.err"; python - "$out/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["path_frac"], d["parity_sample"]["equal"])
print({k: (v["ms_per_launch"], v["launches"]) for k, v in d["kernels"].items() if v["launches"]})
PY
fi
