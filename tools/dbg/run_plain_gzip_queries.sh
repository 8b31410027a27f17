#!/bin/bash
# three COUNT(*) queries on a plain-gzip VCF through the mini host: the first one inflates (serial decoder), the others find the bytes in the pool
cd "$(dirname "$0")/../.."
python3 - <<PY
import sys, gzip; sys.path.insert(0,'tests')
import test_plain_gzip as T
open('/tmp/pg.vcf.gz','wb').write(gzip.compress(T._text(300000, seed=5), 6))
PY
LIB=$(python3 -c "import duckhts_amd; print(duckhts_amd.LIB_PATH)")
tests/minihost/minihost "$LIB" read_bcf /tmp/pg.vcf.gz -t 1 -r 3 -p 0 2>&1 | grep "^RUN\|^OK\|ERROR"
DHTS_FILE_CACHE=0 tests/minihost/minihost "$LIB" read_bcf /tmp/pg.vcf.gz -t 1 -r 2 -p 0 2>&1 | grep "^RUN\|^OK\|ERROR"
