#!/bin/bash
# debug aid: the operator on a multi-GB file with and without the resident-file cache, stage timings on stderr
set -e
N=${1:-30000000}
python - <<PY
import sys; sys.path.insert(0, ".")
from duckhts_amd import synth
synth.bam_segment($N, seed=42)[0].tofile("/tmp/big.bam")
PY
ls -la /tmp/big.bam
for c in 0 1; do
  echo "== DHTS_FILE_CACHE=$c"
  DHTS_FILE_CACHE=$c DHTS_TRACE=1 DHTS_THREADS=8 tests/minihost/minihost duckhts_amd/libduckhts_amd.so read_bam /tmp/big.bam -t 8 -r 3 2>&1 | cut -c1-400
done
