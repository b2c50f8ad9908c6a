#!/bin/bash
# DELTA_BYTE_ARRAY kernel times of variant libraries: tools/dba_variants.sh lib1.so lib2.so ...
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in "$@"; do
  echo "== $lib"
  rm -rf gpurun_out/prof_var
  P4S_BENCH_DUMP_AUX=1 P4S_GPU_LIB=$PWD/$lib timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_var -o p -- python3 bench.py --encodings strings --only delta_byte_array --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-e2e > /dev/null 2> gpurun_out/var.err
  python3 tools/prof_summary.py gpurun_out/prof_var.txt gpurun_out/prof_var 2>/dev/null | grep "k_dba" | grep "^[0-9]*,[0-9]*,[0-9.]*,[0-9.]*,[a-z]" | cut -c1-70
  grep "cycles" gpurun_out/var.err || true
done
