#!/bin/bash
# Times the headline DELTA workload with the experiment builds under exp_build/.
set -e -o pipefail
for lib in "" exp_build/lib_delta_FILLONLY.so exp_build/lib_delta_NODECODE.so exp_build/lib_delta_NOSTORE.so "$@"; do
  if [ -z "$lib" ]; then unset P4S_GPU_LIB; else export P4S_GPU_LIB=$PWD/$lib; fi
  timeout -k 10 300 python3 bench.py --encodings none --no-cpu-baseline --no-verify --no-e2e > gpurun_out/exp_delta.json 2> gpurun_out/exp_delta.err
  python3 -c "
import json
r=json.load(open('gpurun_out/exp_delta.json'))
print('${lib:-base}', r['roofline']['avg_kernel_ms'], 'ms')
"
done
