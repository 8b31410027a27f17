#!/bin/bash
# A/B of environment settings (GPU box): one short bench per "NAME=VALUE[,NAME=VALUE]" argument ("-" = defaults)
out="gpurun_out/ab/env"; mkdir -p "$out"
for spec in "$@"; do
  tag=$(echo "$spec" | tr ',=' '__')
  ( if [ "$spec" != "-" ]; then for kv in ${spec//,/ }; do export "$kv"; done; fi
    python bench.py --steps 3 --warmup 1 --no-extra-configs --no-operator --no-cpu-baseline --no-parity-sample > "$out/$tag.json" 2> "$out/$tag.err" ) || { tail -3 "$out/$tag.err"; continue; }
  python - "$out/$tag.json" "$tag" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: v["ms_per_launch"] for k, v in d["kernels"].items() if v["launches"]})
PY
done
