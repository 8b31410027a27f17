#!/bin/bash
# build_variant.sh NAME [-DMACRO=VALUE ...] -- the library with extra macros as build/libNAME.so (A/B runs: DHTS_LIB=build/libNAME.so, tools/dbg/time_*.py)
set -euo pipefail
root="$(cd "$(dirname "$0")/../.." && pwd)"; name="$1"; shift
mkdir -p "$root/build"
cd "$root/duckhts_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Xclang -target-feature -Xclang +unaligned-ds-access "$@" -o "$root/build/lib$name.so" dhts_api.hip duckdb_ext.cpp duckdb_tools.cpp bcf_header.cpp 2>&1 | grep -v "unaligned-ds-access" || true
ls -la "$root/build/lib$name.so"
