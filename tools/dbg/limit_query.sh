#!/bin/bash
# LIMIT 10-like query (the host stops after one chunk) on the 9.6 GB file: the context must not read the rest of the file before it closes
set -e
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
from duckhts_amd import synth
synth.bam_segment(92_000_000, seed=42)[0].tofile("/tmp/big.bam")
PY
H=tests/minihost/minihost; L=duckhts_amd/libduckhts_amd.so
echo "== read_bam ... LIMIT 10 (first chunk only), file read every query, 4 queries"
DHTS_FILE_CACHE=0 $H $L read_bam /tmp/big.bam -l 10 -r 4 | grep -E "^RUN|^OK"
echo "== the same with the file cache on (the partial file must not be taken for the whole one)"
$H $L read_bam /tmp/big.bam -l 10 -r 3 | grep -E "^RUN|^OK"
echo "== and a full scan afterwards in the same process shape"
DHTS_THREADS=8 $H $L read_bam /tmp/big.bam -t 8 -r 2 | grep -E "^RUN|^OK"
rm -f /tmp/big.bam
