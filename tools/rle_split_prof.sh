#!/bin/bash
# Per-kernel times of one p4s_decode_rle call over a batch of few pages (the split passes): rocprofv3 --kernel-trace --stats.
# usage: tools/rle_split_prof.sh <width> <pages> [structure]
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/split_prof; rm -rf "$out"; mkdir -p "$out"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out" -o r -- python3 tools/rle_probe.py "$1" "$2" "${3:-alt}" > "$out/run.log" 2>&1
cat "$out/run.log" | tail -2
python3 tools/prof_summary.py "$out/summary.txt" "$out" > /dev/null
grep "p4s::" "$out/summary.txt" | cut -c1-120
