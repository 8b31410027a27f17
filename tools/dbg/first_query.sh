#!/bin/bash
# first-query cost against the phase-A look-ahead (scratch size): fresh process each time, 9.6 GB file, all 13 columns
set -e
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
from duckhts_amd import synth
synth.bam_segment(92_000_000, seed=42)[0].tofile("/tmp/big.bam")
PY
H=tests/minihost/minihost; L=duckhts_amd/libduckhts_amd.so
for sb in 196608 98304 49152; do for rep in 1 2; do
  echo "== DHTS_SUPER_BLOCKS=$sb (fresh process $rep)"
  DHTS_THREADS=8 DHTS_TRACE=1 DHTS_SUPER_BLOCKS=$sb $H $L read_bam /tmp/big.bam -t 8 -r 3 2>&1 | grep -E "^RUN|hipMalloc" | tr '\n' ' '; echo
done; done
rm -f /tmp/big.bam
