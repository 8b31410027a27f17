python - <<'PY'
import sys, os; sys.path.insert(0,'.')
from duckhts_amd import synth
synth.bam_segment(8000000, seed=42)[0].tofile('/tmp/s8.bam')
PY
for thr in 1 8; do for p in "1,3,4" "0,1,2,3,4,5,6,7,8,9,10,11,12"; do echo "== threads $thr proj $p"; DHTS_TRACE=1 DHTS_THREADS=$thr timeout 120 tests/minihost/minihost duckhts_amd/libduckhts_amd.so read_bam /tmp/s8.bam -t $thr -r 3 -p $p 2>&1 | grep -v "^OK"; done; done
