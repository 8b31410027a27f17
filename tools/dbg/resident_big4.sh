#!/bin/bash
# why does every other resident query take 0.18 s longer?  producer trace of six queries
set -e
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
from duckhts_amd import synth
synth.bam_segment(92_000_000, seed=42)[0].tofile("/tmp/big.bam")
PY
H=tests/minihost/minihost; L=duckhts_amd/libduckhts_amd.so
DHTS_THREADS=8 DHTS_TRACE=1 $H $L read_bam /tmp/big.bam -t 8 -r 6 2>&1 | grep -E "^RUN|producer|bind|hipMalloc"
rm -f /tmp/big.bam
