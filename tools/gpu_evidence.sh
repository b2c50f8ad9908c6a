#!/bin/bash
# Runs on the GPU box: GPU parity tests, the default bench line, the rocprofv3 kernel
# summary of the same command and the two HBM-traffic counter passes.  Output under
# gpurun_out/$1 (default "ev").  Every GPU step has its own time limit and the script
# stops at the first failure.
set -e -o pipefail
tag=${1:-ev}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
echo "== tests"; timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -40 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
echo "== bench"; timeout -k 10 400 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
cat "$out/bench.json"
echo "== rocprof stats"; timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out/prof" -o r -- python3 bench.py --no-cpu-baseline --no-e2e --rle-widths 1,12,20 > "$out/prof.log" 2>&1 || { tail -20 "$out/prof.log"; exit 1; }
echo "== pmc fetch"; timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-e2e --encodings none > "$out/pmc_fetch.log" 2>&1 || { tail -20 "$out/pmc_fetch.log"; exit 1; }
echo "== pmc write"; timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-e2e --encodings none > "$out/pmc_write.log" 2>&1 || { tail -20 "$out/pmc_write.log"; exit 1; }
find "$out" -name '*.csv' | head -20
echo done
