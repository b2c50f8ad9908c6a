#!/bin/bash
# Times the headline DELTA workload with the default library and with the experiment builds given as arguments
# (paths under exp_build/); the DELTA parity tests run first against each.
set -e -o pipefail
mkdir -p gpurun_out
for lib in "" "$@"; do
  if [ -z "$lib" ]; then unset P4S_GPU_LIB; else export P4S_GPU_LIB=$PWD/$lib; fi
  timeout -k 10 300 python3 -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_delta_bss_plain.py tests/test_gpu_strings.py "tests/test_gpu_full_size.py::test_config2_delta_int64_256m_values" > gpurun_out/exp_delta_tests.log 2>&1 || { tail -30 gpurun_out/exp_delta_tests.log; exit 1; }
  tail -1 gpurun_out/exp_delta_tests.log
  timeout -k 10 300 python3 bench.py --encodings none --no-cpu-baseline --no-verify --no-e2e > gpurun_out/exp_delta.json 2> gpurun_out/exp_delta.err
  python3 -c "
import json
r=json.load(open('gpurun_out/exp_delta.json'))
print('${lib:-base}', r['roofline']['avg_kernel_ms'], 'ms')
"
done
