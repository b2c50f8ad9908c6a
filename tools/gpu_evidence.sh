#!/bin/bash
# Runs on the GPU box: GPU parity tests, the default bench line, the rocprofv3 kernel summary of the same command
# and the per-entry HBM-traffic counter passes (tools/pmc_traffic.sh).  Output under gpurun_out/$1 (default "ev").
# Every GPU step has its own time limit and the script stops at the first failure.
set -e -o pipefail
tag=${1:-ev}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
echo "== tests"; timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -40 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
echo "== bench"; timeout -k 10 600 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
python3 tools/show_enc.py "$out/bench.json"
echo "== rocprof stats"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$out/prof" -o r -- python3 bench.py > "$out/prof.log" 2>&1 || { tail -20 "$out/prof.log"; exit 1; }
python3 tools/prof_summary.py "$out/kernels.txt" "$out/prof" > /dev/null
rm -rf "$out/prof"            # (the summary is what is kept: the box's output directory may not exceed 64 MiB)
# the same with nothing but the headline workload, so that the headline kernel's average is over its own launches
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/prof_headline" -o r -- python3 bench.py --encodings none > "$out/prof_headline.log" 2>&1 || { tail -20 "$out/prof_headline.log"; exit 1; }
python3 tools/prof_summary.py "$out/kernels_headline.txt" "$out/prof_headline" > /dev/null
rm -rf "$out/prof_headline"
if [ -n "$SKIP_TRAFFIC" ]; then echo done; exit 0; fi      # (the counter passes take minutes: they can run in a call of their own)
echo "== traffic"; bash tools/pmc_traffic.sh "$tag/traffic" > "$out/traffic.log" 2>&1 || { tail -30 "$out/traffic.log"; exit 1; }
tail -5 "$out/traffic.log"
echo done
