#!/bin/bash
# cache-off queries of a 9.6 GB file, four in one process, with the producer trace
set -e
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
from duckhts_amd import synth
synth.bam_segment(92_000_000, seed=42)[0].tofile("/tmp/big.bam")
PY
H=tests/minihost/minihost; L=duckhts_amd/libduckhts_amd.so
for k in 1 2; do
echo "== cache off, pass $k"
DHTS_THREADS=8 DHTS_FILE_CACHE=0 DHTS_TRACE=1 $H $L read_bam /tmp/big.bam -t 8 -r 4 2>&1 | grep -E "^RUN|producer"
done
rm -f /tmp/big.bam
