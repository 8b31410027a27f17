#!/bin/bash
# LDS staging budget of the lane-per-line VCF text encoder: COUNT(*) and four columns on the ClinVar-shaped file and on a 10-sample cohort file,
# file resident, for the automatic budget, the old fixed 40 KiB, 16 KiB and no staging.   tools/dbg/vcf_lds_sweep.sh [records]
cd "$(dirname "$0")/../.."
N=${1:-2000000}
python3 - <<PY
import sys; sys.path.insert(0,'tools'); sys.argv=['x']
import bench_vcf_text as b
b.generate('/tmp/cv.vcf.gz', $N)
b.generate_samples_shape('/tmp/s10.vcf.gz', $N // 4, 10)
PY
LIB=$(python3 -c "import duckhts_amd; print(duckhts_amd.LIB_PATH)")
for f in /tmp/cv.vcf.gz /tmp/s10.vcf.gz; do
  for cfg in "auto" "DHTS_VCF_LDS=40960" "DHTS_VCF_LDS=16384" "DHTS_VCF_STAGE=0"; do
    for proj in 0 0,1,3,4; do
      if [ "$cfg" = auto ]; then E=""; else E="$cfg"; fi
      R=$(env DHTS_FILE_CACHE=1 DHTS_THREADS=1 $E tests/minihost/minihost "$LIB" read_bcf $f -t 1 -r 5 -p $proj 2>&1 | grep "^RUN" | awk '{print $3}' | sed 's/seconds=//' | sort -n | head -2 | tail -1)
      echo "$(basename $f) $cfg proj=$proj second-best-of-5 $R s"
    done
  done
done
