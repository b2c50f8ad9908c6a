#!/usr/bin/env python3
"""Builds a variant of the HIP library under exp_build/ for A/B experiments:
   tools/build_variant.py <name> [-DFOO[=x] ...] [--only src.hip,...]   ->  exp_build/libp4s_gpu_<name>.so"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g      # noqa: E402

name = sys.argv[1]
defines = [a[2:] for a in sys.argv[2:] if a.startswith("-D")]
if any(d.startswith("P4S_EXP_") for d in defines):
    defines.append("P4S_EXPERIMENTS")            # the experiment switches only compile together with this one
only = None
for a in sys.argv[2:]:
    if a.startswith("--only="):
        only = a[7:].split(",")
os.makedirs(os.path.join(ROOT, "exp_build"), exist_ok=True)
g.build_hip(out=os.path.join(ROOT, "exp_build", "libp4s_gpu_%s.so" % name), defines=defines, only=only,
            unity="--unity" in sys.argv)
