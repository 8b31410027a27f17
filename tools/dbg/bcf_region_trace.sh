#!/bin/bash
# where a small region query of read_bcf on bgzipped VCF text spends its time (DHTS_TRACE), 4 queries in one process
set -e
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import bench_vcf_text as B, duckhts_amd, ctypes as C, numpy as np
B.generate("/tmp/cv.vcf.gz", 4352930)
L = duckhts_amd.lib(); L.dhts_bcf_build_index.restype = C.c_int64; L.dhts_bcf_build_index.argtypes = [C.c_void_p, C.c_int]
ctx = duckhts_amd.Context(0); ctx.open("/tmp/cv.vcf.gz"); ctx.bgzf_index(); duckhts_amd.BcfScan(ctx)
k = L.dhts_bcf_build_index(ctx.h, 0); raw = np.zeros(k, np.uint8); L.dhts_bam_index_bytes(ctx.h, raw.ctypes.data, k)
open("/tmp/cv.vcf.gz.tbi", "wb").write(ctx.bgzf_compress(raw.tobytes())); ctx.close()
PY
H=tests/minihost/minihost; L=duckhts_amd/libduckhts_amd.so
DHTS_TRACE=1 DHTS_FILE_CACHE=0 $H $L read_bcf /tmp/cv.vcf.gz -t 1 -r 4 -p 0 -n region=1:1000000-1100000 2>&1 | grep -E "^RUN|dhts"
rm -f /tmp/cv.vcf.gz /tmp/cv.vcf.gz.tbi
