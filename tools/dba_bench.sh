#!/bin/bash
# DELTA_BYTE_ARRAY / DELTA_LENGTH_BYTE_ARRAY bench entries only, with the kernels' rocprofv3 stats.  usage: tools/dba_bench.sh <tag>
set -e
tag=${1:-x}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 240 python3 bench.py --encodings strings --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/dba_${tag}.json 2> gpurun_out/dba_${tag}.err
python3 - <<PY
import json
d = json.loads(open("gpurun_out/dba_${tag}.json").read().strip().splitlines()[-1])
for k, v in d.get("encodings", {}).items():
    print(k, v.get("ms"), v.get("GB/s"), v.get("frac_of_8TBs", v.get("frac")))
PY
rm -rf gpurun_out/prof_${tag}
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag} -o p -- python3 bench.py --encodings strings --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > /dev/null 2> gpurun_out/dba_${tag}_prof.err
python3 tools/prof_summary.py gpurun_out/prof_${tag}.txt gpurun_out/prof_${tag} 2>/dev/null | grep -i "dba\|length_byte\|delta_byte" | head -20 || true
find gpurun_out/prof_${tag} -name "*.db" -delete 2>/dev/null || true
