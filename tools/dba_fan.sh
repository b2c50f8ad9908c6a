#!/bin/bash
# k_dba_assemble time against the number of workgroups per page: tools/dba_fan.sh 1024 4096 16384 65536
set -e
export TMPDIR=/tmp
for f in "$@"; do
  rm -rf gpurun_out/prof_fan
  P4S_DBA_FAN=$f timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fan -o p -- python3 bench.py --encodings strings --only delta_byte_array --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > /dev/null 2> gpurun_out/fan.err
  echo "== P4S_DBA_FAN=$f"
  python3 tools/prof_summary.py gpurun_out/prof_fan.txt gpurun_out/prof_fan 2>/dev/null | grep "k_dba_assemble" | tail -1 | cut -c1-60
done
