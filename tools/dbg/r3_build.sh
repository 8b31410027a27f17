#!/bin/bash
# round-3 helper (CPU container): host simulation of the wave kernel, library + -DHW_DIAG variant
set -e
cd "$(dirname "$0")/../.."
python -m pytest tests/test_hostsim.py -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -v "unaligned-ds-access" | tail -3
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DHW_DIAG ${EXTRA_DEFS:-} -o build/lib_diag.so duckhts_amd/csrc/dhts_api.hip duckhts_amd/csrc/duckdb_ext.cpp duckhts_amd/csrc/duckdb_tools.cpp duckhts_amd/csrc/bcf_header.cpp 2>&1 | grep -E "error" | head -3 || true
echo build done
