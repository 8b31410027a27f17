#!/bin/bash
# why is the resident-file query of a multi-GB file slower than the cold one?  (bench.py operator.file_resident, 9.58 GB: 1.72 s vs 1.04 s)
set -e
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
from duckhts_amd import synth
synth.bam_segment(92_000_000, seed=42)[0].tofile("/tmp/big.bam")
PY
ls -la /tmp/big.bam
H=tests/minihost/minihost; L=duckhts_amd/libduckhts_amd.so
for sb in "" 524288; do
  echo "== DHTS_SUPER_BLOCKS=$sb cache on"
  DHTS_THREADS=8 DHTS_SUPER_BLOCKS=$sb $H $L read_bam /tmp/big.bam -t 8 -r 4 | grep -E "^RUN|^OK"
done
echo "== cache off"
DHTS_THREADS=8 DHTS_FILE_CACHE=0 $H $L read_bam /tmp/big.bam -t 8 -r 3 | grep -E "^RUN|^OK"
echo "== trace, cache on, third query"
DHTS_THREADS=8 DHTS_TRACE=1 $H $L read_bam /tmp/big.bam -t 8 -r 3 2>&1 | grep -v "^RUN\|^OK" | tail -40
rm -f /tmp/big.bam
