#!/bin/bash
# debug aid: per-query fixed cost of the operator: the 13 KB golden file, whole and with a region, stage timings on stderr
for args in "" "-n region=CHROMOSOME_I:1-5000"; do
  echo "== $args"
  DHTS_TRACE=1 tests/minihost/minihost duckhts_amd/libduckhts_amd.so read_bam tests/golden/range.bam $args -r 4 2>&1 | cut -c1-330
done
