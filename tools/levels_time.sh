#!/bin/bash
# Level streams with 10 % nulls through p4s_decode_levels: the tree's library, then timing builds (cycle counters of
# wave 0 and of an expanding wave).  usage: tools/levels_time.sh [lib ...]
set -e -o pipefail
export P4S_PROBE_LEVELS=1
timeout -k 10 200 python3 tools/rle_probe.py 1 2048 nulls
for lib in "$@"; do
  P4S_PROBE_AUX=1 P4S_GPU_LIB=$PWD/$lib timeout -k 10 200 python3 tools/rle_probe.py 1 2048 nulls
done
