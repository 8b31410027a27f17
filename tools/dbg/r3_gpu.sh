#!/bin/bash
# round-3 helper (GPU box): BAM parity tests, per-phase cycle split, phase-A timing, optional SQ counters (PMC=1); output under gpurun_out/r3/<tag>
tag="${1:-x}"; out="gpurun_out/r3/$tag"; mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_gpu_bam.py -x -q -m gpu > "$out/tests.txt" 2>&1; tail -2 "$out/tests.txt"
DHTS_LIB=build/lib_diag.so timeout -k 10 200 python tools/dbg/hw_diag.py > "$out/hw_diag.txt" 2>&1; cat "$out/hw_diag.txt"
DHTS_PHASE_A=wave timeout -k 10 200 python tools/dbg/time_huff.py > "$out/time_huff.txt" 2>&1; tail -2 "$out/time_huff.txt"
if [ -n "${PMC:-}" ]; then DHTS_PHASE_A=wave timeout -k 10 600 bash tools/dbg/pmc_sq.sh "$out/pmc_sq" > "$out/pmc_sq.txt" 2>&1; tail -1 "$out/pmc_sq.txt"; fi
