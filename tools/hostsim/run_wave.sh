#!/bin/bash
# Host build of the wave-per-block Huffman kernel (bgzf_huff_wave.hip) under ASAN/UBSAN, cross-checked against the host build of
# the lane-per-block kernel text and against every block's CRC32/ISIZE trailer.
#   tools/hostsim/run_wave.sh [--flip N seed] file.bam [more.bgzf ...]
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"; root="$(cd "$here/../.." && pwd)"
gen="$here/_gen"; mkdir -p "$gen"
src="$root/duckhts_amd/csrc/bgzf_inflate.hip"
a=$(grep -n '^// phase A$' "$src" | head -1 | cut -d: -f1)
b=$(grep -n '^// phase B$' "$src" | head -1 | cut -d: -f1)
sed -n "$((a + 2)),$((b - 2))p" "$src" \
  | sed 's/extern __shared__ __attribute__((aligned(16))) uint8_t smem\[\];/uint8_t *smem = g_smem;/' > "$gen/phaseA_extract.inc"
if [ ! -x "$gen/sim_wave" ] || [ "$here/sim_wave.cpp" -nt "$gen/sim_wave" ] || [ "$root/duckhts_amd/csrc/bgzf_huff_wave.hip" -nt "$gen/sim_wave" ] || [ "$src" -nt "$gen/sim_wave" ]; then
  g++ ${SIM_OPT:--O1} -g -std=c++17 -fsanitize=address,undefined -fno-sanitize=shift-base -fno-sanitize-recover=undefined -I"$gen" -o "$gen/sim_wave" "$here/sim_wave.cpp"
fi
"$gen/sim_wave" "$@"
