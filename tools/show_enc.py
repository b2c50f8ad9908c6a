#!/usr/bin/env python3
"""Prints the per-encoding entries of a bench.py JSON line: name, ms, GB/s, fraction of the HBM peak."""
import json
import sys

r = json.load(open(sys.argv[1]))
for k, e in r.get("encodings", {}).items():
    if "ms" not in e:
        continue
    print("%-28s %8.3f ms  %s" % (k, e["ms"], " ".join("%s=%s" % (n, e[n]) for n in ("GB/s", "hbm_frac") if n in e)))
