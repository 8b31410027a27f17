#!/bin/bash
# a cohort-shaped (many samples) bgzipped VCF through the mini host: tools/dbg/run_cohort.sh [samples] [records] [projection]
cd "$(dirname "$0")/../.."
NS=${1:-2504}; NR=${2:-2000}; PROJ=${3:-0}
python3 - <<PY
import sys; sys.path.insert(0,'tools'); sys.argv=['x']
import bench_vcf_text as b
b.generate_samples_shape('/tmp/coh.vcf.gz', $NR, $NS)
PY
LIB=$(python3 -c "import duckhts_amd; print(duckhts_amd.LIB_PATH)")
DHTS_TRACE=1 tests/minihost/minihost "$LIB" read_bcf /tmp/coh.vcf.gz -t 1 -r 3 -p "$PROJ" > /tmp/coh.out 2>&1
echo "rc=$?"; tail -12 /tmp/coh.out | cut -c1-300
