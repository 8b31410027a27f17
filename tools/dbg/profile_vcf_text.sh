#!/bin/bash
# Kernel trace of read_bcf on the gnomAD-shaped bgzipped VCF text (COUNT(*) through the mini host, file resident).
#   tools/dbg/profile_vcf_text.sh [records] [projection]     -> gpurun_out/vcfprof/
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
N=${1:-60000}; PROJ=${2:-0}
OUT=$ROOT/gpurun_out/vcfprof; mkdir -p "$OUT"
F=/tmp/gn_$N.vcf.bgz
[ -s "$F" ] || python3 -c "
import sys; sys.path.insert(0, '$ROOT/tools'); sys.argv=['x']
import bench_vcf_text as b; b.generate_gnomad_shape('$F', $N)"
LIB=$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); import duckhts_amd; print(duckhts_amd.LIB_PATH)")
export DHTS_THREADS=1 DHTS_FILE_CACHE=1
DHTS_TRACE=1 "$ROOT/tests/minihost/minihost" "$LIB" read_bcf "$F" -t 1 -r 4 -p "$PROJ" > "$OUT/plain_p$PROJ.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o gn_p$PROJ --output-format csv -- "$ROOT/tests/minihost/minihost" "$LIB" read_bcf "$F" -t 1 -r 4 -p "$PROJ" > "$OUT/prof_p$PROJ.txt" 2>&1
python3 - "$OUT/gn_p${PROJ}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print("%-48s calls %6s total %10.3f ms avg %9.1f us  %5s%%" % (r["Name"][:48], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
