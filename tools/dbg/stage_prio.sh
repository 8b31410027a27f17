#!/bin/bash
# which queue class should the staging (host -> device) streams use?  file read every query, 9.6 GB, all 13 columns
set -e
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
from duckhts_amd import synth
synth.bam_segment(92_000_000, seed=42)[0].tofile("/tmp/big.bam")
PY
H=tests/minihost/minihost; L=duckhts_amd/libduckhts_amd.so
for pr in 0 1 2; do
  echo "== DHTS_STAGE_PRIO=$pr, cache off"
  DHTS_THREADS=8 DHTS_FILE_CACHE=0 DHTS_STAGE_PRIO=$pr $H $L read_bam /tmp/big.bam -t 8 -r 5 | grep -E "^RUN" | tr '\n' ' '; echo
done
rm -f /tmp/big.bam
