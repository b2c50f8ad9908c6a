#!/bin/bash
# String-encoding parity tests and timing with the library in the tree, then phase cycle counts from a
# -DP4S_EXP_TIMING build given as argument.
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_strings.py tests/test_gpu_host.py > gpurun_out/exp_str_tests.log 2>&1 || { tail -30 gpurun_out/exp_str_tests.log; exit 1; }
tail -1 gpurun_out/exp_str_tests.log
timeout -k 10 300 python3 bench.py --encodings strings --no-cpu-baseline --no-e2e > gpurun_out/exp_str.json 2> gpurun_out/exp_str.err
python3 -c "
import json
r=json.load(open('gpurun_out/exp_str.json'))
print(' '.join('%s=%.3fms' % (k, e['ms']) for k, e in r['encodings'].items()))
"
for lib in "$@"; do
  P4S_BENCH_DUMP_AUX=1 P4S_GPU_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --encodings strings --no-cpu-baseline --no-verify --no-e2e 2>&1 >/dev/null | grep "aux cycles" || true
done
