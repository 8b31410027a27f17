#!/bin/bash
# Config 3 (read_bcf) under the profiler, on the GPU box: rocprofv3 kernel trace + stats of tools/bench_bcf.py (count(*) and all 111 columns),
# then the FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes), reduced per kernel by tools/pmc_traffic.py.
#   tools/profile_bcf.sh <outdir under gpurun_out> <version tag>
set -uo pipefail
out="${1:-gpurun_out/prof_bcf}"; ver="${2:-v1}"; repo="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p "$repo/$out"
cd /tmp && export TMPDIR=/tmp
q="--queries count,all --cpu-sample-records 0"
echo "kernel trace"; rocprofv3 --kernel-trace --stats -d "$repo/$out/kt" -o kt --output-format csv -- python3 "$repo/tools/bench_bcf.py" $q > "$repo/$out/bcf_${ver}_bench_under_rocprof.jsonl" 2> "$repo/$out/kt.err"
echo "pmc fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$repo/$out/pf" -o pf --output-format csv -- python3 "$repo/tools/bench_bcf.py" --steps 1 --warmup 0 $q > "$repo/$out/pf.jsonl" 2> "$repo/$out/pf.err"
echo "pmc write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$repo/$out/pw" -o pw --output-format csv -- python3 "$repo/tools/bench_bcf.py" --steps 1 --warmup 0 $q > "$repo/$out/pw.jsonl" 2> "$repo/$out/pw.err"
python3 "$repo/tools/pmc_traffic.py" "$repo/$out/pf" "$repo/$out/pw" "$repo/$out/bcf_pmc_traffic_${ver}.json" cmd="tools/bench_bcf.py --steps 1 --warmup 0 --queries count,all (1.05 GB BCF, 28,658 blocks, one pass per query)" || true
kt=$(find "$repo/$out/kt" -name "*kernel_trace.csv" | head -1)
[ -n "$kt" ] && python3 "$repo/tools/dbg/kernel_launches.py" "$kt" "tools/bench_bcf.py --queries count,all under rocprofv3 --kernel-trace ($ver)" > "$repo/$out/bcf_${ver}_kernel_launches.txt"
st=$(find "$repo/$out/kt" -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp "$st" "$repo/$out/bcf_${ver}_kernel_stats.csv"
for p in pf pw; do c=$(find "$repo/$out/$p" -name "*counter_collection.csv" | head -1); [ -n "$c" ] && cp "$c" "$repo/$out/bcf_pmc_${ver}_${p}_counter_collection.csv"; done
find "$repo/$out" -name "*kernel_trace.csv" -size +4M -delete
find "$repo/$out" -path "*/p[fw]/*" -name "*.csv" -size +8M -delete
ls -la "$repo/$out" | head -30
