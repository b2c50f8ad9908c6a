#!/bin/bash
# Runs GPU parity tests against the bounds-checked build (-DP4S_DEBUG_BOUNDS) of the library:
# accesses outside the call's buffers are reported by the test helpers instead of executed.
#   tools/dbg_bounds.sh <log tag> <pytest args...>
set -u
tag=$1; shift
mkdir -p gpurun_out
export P4S_GPU_LIB=$PWD/exp_build/libp4s_gpu_dbg.so
timeout -k 10 240 python3 -u -m pytest -x -q --timeout 60 --timeout-method thread "$@" > gpurun_out/dbg_$tag.log 2>&1
rc=$?
grep -v "^$" gpurun_out/dbg_$tag.log | tail -25
exit $rc
