#!/bin/bash
# packed SEQ read-back on / off, 8 and 16 fill threads, resident 9.6 GB file
set -e
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
from duckhts_amd import synth
synth.bam_segment(92_000_000, seed=42)[0].tofile("/tmp/big.bam")
PY
H=tests/minihost/minihost; L=duckhts_amd/libduckhts_amd.so
for thr in 8 16; do for pk in 1 0; do
  echo "== threads $thr packed $pk (cache on)"
  DHTS_THREADS=$thr DHTS_SEQ_PACKED=$pk $H $L read_bam /tmp/big.bam -t $thr -r 5 | grep -E "^RUN" | tr '\n' ' '; echo
done; done
echo "== threads 16 packed 1 cache off"
DHTS_THREADS=16 DHTS_FILE_CACHE=0 $H $L read_bam /tmp/big.bam -t 16 -r 4 | grep -E "^RUN" | tr '\n' ' '; echo
rm -f /tmp/big.bam
