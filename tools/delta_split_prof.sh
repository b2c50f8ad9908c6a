#!/bin/bash
# Per-kernel times of p4s_decode_delta_binary_packed over a batch of few pages: rocprofv3 --kernel-trace --stats.
# usage: tools/delta_split_prof.sh <pages>
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/delta_prof; rm -rf "$out"; mkdir -p "$out"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out" -o r -- python3 tools/delta_probe.py "$1" > "$out/run.log" 2>&1
tail -2 "$out/run.log"
python3 tools/prof_summary.py "$out/summary.txt" "$out" > /dev/null
grep "p4s::" "$out/summary.txt" | cut -c1-110
