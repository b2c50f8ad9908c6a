#!/bin/bash
# Instruction-mix counters for the headline DELTA kernel (one rocprofv3 --pmc pass, SQ block only).
# usage: tools/pmc_sq.sh <tag> [library]
set -e -o pipefail
tag=${1:-sq}; lib=$2
out=gpurun_out/$tag; mkdir -p "$out"
if [ -n "$lib" ]; then export P4S_GPU_LIB=$PWD/$lib; fi
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d "$out/pmc" -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-e2e --encodings none > "$out/pmc.log" 2>&1 || { tail -20 "$out/pmc.log"; exit 1; }
python3 tools/prof_summary.py "$out/summary.txt" "$out/pmc"
grep -i "delta" "$out/summary.txt" || cat "$out/summary.txt"
