#!/bin/bash
# A/B timing of experimental builds of the library (same process layout, back to back)
for lib in "$@"; do
  echo "== $lib"
  P4S_GPU_LIB=$PWD/exp_build/$lib timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['value'])"
done
