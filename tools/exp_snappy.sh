#!/bin/bash
# Times the Snappy workloads with the library in the tree and with each experiment library given as argument
# (experiment libraries may produce wrong bytes: no verification for them).
set -e -o pipefail
mkdir -p gpurun_out
for lib in "" "$@"; do
  extra=""
  if [ -n "$lib" ]; then export P4S_GPU_LIB=$PWD/$lib; extra="--no-verify"; fi
  timeout -k 10 300 python3 bench.py --encodings snappy --no-cpu-baseline --no-e2e $extra > gpurun_out/exp_sn.json 2> gpurun_out/exp_sn.err
  python3 -c "
import json
r=json.load(open('gpurun_out/exp_sn.json'))
print('${lib:-tree}', ' '.join('%s=%.3fms' % (k, e['ms']) for k, e in r['encodings'].items()))
"
done
