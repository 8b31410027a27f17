#!/bin/bash
# The round's judged profiles (run on the GPU box): rocprofv3 kernel trace + stats of the bench command, then the two PMC traffic passes.
#   tools/dbg/profile_round.sh <outdir under gpurun_out>
set -uo pipefail
out="${1:-gpurun_out/prof_r02}"; repo="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p "$repo/$out"
cd /tmp && export TMPDIR=/tmp
echo "kernel trace"; rocprofv3 --kernel-trace --stats -d "$repo/$out/kt" -o kt --output-format csv -- python3 "$repo/bench.py" --no-operator --no-parity-sample --no-cpu-baseline > "$repo/$out/bench_under_rocprof.json" 2> "$repo/$out/kt.err"
echo "pmc fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$repo/$out/pf" -o pf --output-format csv -- python3 "$repo/bench.py" --target-gb 3.3 --steps 1 --warmup 0 --no-operator --no-parity-sample --no-cpu-baseline > "$repo/$out/pf.json" 2> "$repo/$out/pf.err"
echo "pmc write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$repo/$out/pw" -o pw --output-format csv -- python3 "$repo/bench.py" --target-gb 3.3 --steps 1 --warmup 0 --no-operator --no-parity-sample --no-cpu-baseline > "$repo/$out/pw.json" 2> "$repo/$out/pw.err"
python3 "$repo/tools/pmc_traffic.py" "$repo/$out/pf" "$repo/$out/pw" "$repo/$out/pmc_traffic_v15.json" cmd="bench.py --target-gb 3.3 --steps 1 --warmup 0" || true
ls -la "$repo/$out" "$repo/$out/kt" | head -30
# keep the merge small: drop the raw kernel traces, keep the stats
find "$repo/$out" -name "*kernel_trace.csv" -size +8M -delete
