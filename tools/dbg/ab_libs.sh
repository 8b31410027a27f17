#!/bin/bash
# A/B of library variants (GPU box): short bench per variant library given as arguments (paths under build/)
out="gpurun_out/ab/var"; mkdir -p "$out"
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  DHTS_LIB="$lib" python bench.py --steps 3 --warmup 1 --no-extra-configs --no-operator --no-cpu-baseline --no-parity-sample > "$out/$tag.json" 2> "$out/$tag.err" || { tail -3 "$out/$tag.err"; continue; }
  python - "$out/$tag.json" "$tag" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["value"], d["ms_per_step"], {k: v["ms_per_launch"] for k, v in d["kernels"].items() if v["launches"]})
PY
done
