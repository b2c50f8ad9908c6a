#!/bin/bash
# Times the string encodings with the library in the tree and with each experiment library given as argument.
set -e -o pipefail
mkdir -p gpurun_out
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export P4S_GPU_LIB=$PWD/$lib; fi
  timeout -k 10 300 python3 bench.py --encodings strings --no-cpu-baseline --no-e2e > gpurun_out/exp_ls.json 2> gpurun_out/exp_ls.err
  python3 -c "
import json,sys
r=json.load(open('gpurun_out/exp_ls.json'))
print('${lib:-tree}', ' '.join('%s=%.3fms' % (k, e['ms']) for k, e in r['encodings'].items()), 'verified', r.get('verified_bit_exact'))
"
done
