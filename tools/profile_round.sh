#!/bin/bash
# The round's judged profiles (run on the GPU box): rocprofv3 kernel trace + stats of the bench command, then the two PMC traffic passes
# (FETCH_SIZE and WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes) AT THE BENCH'S OWN SIZE, reduced by tools/pmc_traffic.py.
#   tools/profile_round.sh <outdir under gpurun_out> <version tag, e.g. v18>
# Copy what is to be judged from <outdir> into profiles/<round>/ afterwards (gpurun_out/ is scratch).
set -uo pipefail
out="${1:-gpurun_out/prof_r03}"; ver="${2:-v18}"; repo="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p "$repo/$out"
cd /tmp && export TMPDIR=/tmp
common="--no-operator --no-parity-sample --no-cpu-baseline --no-extra-configs"
echo "kernel trace"; rocprofv3 --kernel-trace --stats -d "$repo/$out/kt" -o kt --output-format csv -- python3 "$repo/bench.py" $common > "$repo/$out/${ver}_bench_under_rocprof.json" 2> "$repo/$out/kt.err"
echo "pmc fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$repo/$out/pf" -o pf --output-format csv -- python3 "$repo/bench.py" --steps 1 --warmup 0 $common > "$repo/$out/pf.json" 2> "$repo/$out/pf.err"
echo "pmc write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$repo/$out/pw" -o pw --output-format csv -- python3 "$repo/bench.py" --steps 1 --warmup 0 $common > "$repo/$out/pw.json" 2> "$repo/$out/pw.err"
python3 "$repo/tools/pmc_traffic.py" "$repo/$out/pf" "$repo/$out/pw" "$repo/$out/pmc_traffic_${ver}.json" cmd="bench.py --steps 1 --warmup 0 (the bench's own 9.58 GB file, 455,655 blocks)" || true
kt=$(find "$repo/$out/kt" -name "*kernel_trace.csv" | head -1)
[ -n "$kt" ] && python3 "$repo/tools/dbg/kernel_launches.py" "$kt" "bench.py under rocprofv3 --kernel-trace ($ver)" > "$repo/$out/${ver}_kernel_launches.txt"
st=$(find "$repo/$out/kt" -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp "$st" "$repo/$out/${ver}_kernel_stats.csv"
for p in pf pw; do c=$(find "$repo/$out/$p" -name "*counter_collection.csv" | head -1); [ -n "$c" ] && cp "$c" "$repo/$out/pmc_${ver}_${p}_counter_collection.csv"; done
# keep the merge small: drop the raw traces
find "$repo/$out" -name "*kernel_trace.csv" -size +4M -delete
find "$repo/$out" -path "*/p[fw]/*" -name "*.csv" -size +8M -delete
ls -la "$repo/$out" | head -30
