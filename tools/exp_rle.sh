#!/bin/bash
# RLE parity tests and timing: the library in the tree first (it holds the working copy's kernels), then each
# experiment build given as argument (P4S_GPU_LIB selects the library).
# usage: tools/exp_rle.sh <widths> [lib ...]
set -e -o pipefail
mkdir -p gpurun_out
widths=${1:-1,12}; shift || true
for lib in "" "$@"; do
  if [ -z "$lib" ]; then unset P4S_GPU_LIB; else export P4S_GPU_LIB=$PWD/$lib; fi
  timeout -k 10 300 python3 -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_rle.py tests/test_gpu_host.py > gpurun_out/exp_rle_tests.log 2>&1 || { tail -30 gpurun_out/exp_rle_tests.log; exit 1; }
  tail -1 gpurun_out/exp_rle_tests.log
  timeout -k 10 300 python3 bench.py --encodings rle,dict --rle-widths $widths --no-cpu-baseline --no-verify --no-e2e --steps 2 --warmup 1 > gpurun_out/exp_rle.json 2> gpurun_out/exp_rle.err
  python3 -c "
import json
r=json.load(open('gpurun_out/exp_rle.json'))
print('${lib:-tree}', ' '.join('%s=%.3fms' % (k, e['ms']) for k, e in r['encodings'].items()))
"
done
