#!/bin/bash
# Times the RLE workload with the current library and the experiment builds under exp_build/
# (P4S_GPU_LIB selects the library).
set -e -o pipefail
widths=${1:-1,12}
for lib in "" exp_build/lib_rle_NOEXPAND.so; do
  if [ -z "$lib" ]; then unset P4S_GPU_LIB; else export P4S_GPU_LIB=$PWD/$lib; fi
  timeout -k 10 300 python3 bench.py --encodings rle --rle-widths $widths --no-cpu-baseline --no-verify --no-e2e --steps 2 --warmup 1 > gpurun_out/exp_rle.json 2> gpurun_out/exp_rle.err
  python3 -c "
import json
r=json.load(open('gpurun_out/exp_rle.json'))
print('${lib:-base}', ' '.join('%s=%.3fms' % (k, e['ms']) for k, e in r['encodings'].items()))
"
done
