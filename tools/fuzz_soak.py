#!/usr/bin/env python3
"""Runs the mutated-input GPU parity tests over many seeds in one process (meant for the bounds-checked build:
P4S_GPU_LIB=exp_build/libp4s_gpu_dbg.so).  Stops at the first seed that fails.

    python3 tools/fuzz_soak.py FIRST_SEED N_SEEDS [snappy]
"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_zz_fuzz as fz      # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
only = sys.argv[3] if len(sys.argv) > 3 else ""        # "snappy": that test alone
t0 = time.time()
for seed in range(first, first + count):
    try:
        if only == "snappy":
            fz.test_snappy_mutated(seed=seed)
            print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
            continue
        fz.test_mutated_string_pages(seed=seed)
        for es in (2, 4):
            fz.test_rle_mutated_streams(es, seed=seed)
        for es in (4, 8):
            fz.test_delta_mutated_pages(es, seed=seed)
        fz.test_plain_byte_array_mutated(seed=seed)
        for es in (4, 8):
            fz.test_dictionary_mutated(es, seed=seed)
        fz.test_levels_mutated(seed=seed)
        fz.test_snappy_mutated(seed=seed)
    except Exception:
        traceback.print_exc()
        print("seed %d FAILED" % seed, flush=True)
        sys.exit(1)
    print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("all %d seeds ok" % count)
