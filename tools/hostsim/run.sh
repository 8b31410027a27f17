#!/bin/bash
# Host build of phase A (bgzf_huff_decode) under ASAN/UBSAN: the kernel text is cut out of bgzf_inflate.hip at build time
# (nothing is duplicated in the tree), every lane runs as an independent call on an exact-size LDS image, and the tokens are
# replayed on the host and checked against each block's CRC32/ISIZE trailer.
#   tools/hostsim/run.sh file.bam [more.bgzf ...]
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"; root="$(cd "$here/../.." && pwd)"
gen="$here/_gen"; mkdir -p "$gen"
src="$root/duckhts_amd/csrc/bgzf_inflate.hip"
a=$(grep -n '^// phase A$' "$src" | head -1 | cut -d: -f1)
b=$(grep -n '^// phase B$' "$src" | head -1 | cut -d: -f1)
sed -n "$((a + 2)),$((b - 2))p" "$src" \
  | sed 's/extern __shared__ __attribute__((aligned(16))) uint8_t smem\[\];/uint8_t *smem = g_smem;/' > "$gen/phaseA_extract.inc"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -I"$gen" -o "$gen/sim_huff" "$here/sim_huff.cpp"
for f in "$@"; do NLO=196 "$gen/sim_huff" "$f"; NLO=288 "$gen/sim_huff" "$f"; done
