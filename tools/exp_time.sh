#!/bin/bash
# Times the headline DELTA workload (kernel only, no parity checks) with each experiment library given.
set -e -o pipefail
mkdir -p gpurun_out
for lib in "$@"; do
  P4S_GPU_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --encodings none --no-cpu-baseline --no-verify --no-e2e > gpurun_out/exp_time.json 2> gpurun_out/exp_time.err
  python3 -c "
import json
r=json.load(open('gpurun_out/exp_time.json'))
print('$lib', r['roofline']['avg_kernel_ms'], 'ms')
"
done
