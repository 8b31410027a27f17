#!/bin/bash
# every tools/soak.py mode for SECS seconds each, one line per mode:  tools/dbg/soak_all.sh [seconds] [first seed] > file
cd "$(dirname "$0")/../.."
SECS=${1:-60}; FIRST=${2:-200000}
for m in "" --corrupt --scans --surface --regions --vcf --vcfregions --bgzip --isize; do
  name=${m#--}; [ -z "$name" ] && name=plain
  out=$(timeout -k 10 $((SECS + 120)) python3 tools/soak.py $m --seeds 1000000 --first $FIRST --seconds $SECS 2>&1 | tail -1)
  printf "%-12s %s\n" "$name" "$out"
  case "$out" in *" 0 with mismatches"*) ;; *) echo "STOP: $name did not end clean"; exit 1;; esac
done
