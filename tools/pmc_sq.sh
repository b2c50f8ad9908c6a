#!/bin/bash
# SQ counters for one bench entry's kernels (one rocprofv3 --pmc pass per counter set, SQ block only); without an
# entry: the headline DELTA kernel.
# usage: tools/pmc_sq.sh <tag> <set: mix|wait|mem|icache> [library|-] [family entry]     e.g. ... - strings delta_byte_array
set -e -o pipefail
tag=${1:-sq}; set_=${2:-mix}; lib=$3; family=${4:-none}; entry=$5
case $set_ in
  mix)  ctrs="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY";;
  wait) ctrs="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM";;
  mem)  ctrs="SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM";;
  icache) ctrs="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_SALU";;
esac
out=gpurun_out/$tag; mkdir -p "$out"
if [ -n "$lib" ] && [ "$lib" != - ]; then export P4S_GPU_LIB=$PWD/$lib; fi
extra=""
if [ -n "$entry" ]; then extra="--only $entry --pages 16 --enc-steps 2"; fi
timeout -s KILL 200 rocprofv3 --pmc $ctrs -d "$out/pmc_$set_" -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-e2e --encodings $family $extra > "$out/pmc_$set_.log" 2>&1 || { tail -20 "$out/pmc_$set_.log"; exit 1; }
python3 tools/prof_summary.py "$out/summary_$set_.txt" "$out/pmc_$set_" > /dev/null
grep "p4s::" "$out/summary_$set_.txt" | cut -c1-110
