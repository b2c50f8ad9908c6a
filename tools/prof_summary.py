#!/usr/bin/env python3
"""Condenses rocprofv3 output (rocpd sqlite `*_results.db`) into the small text
summaries kept under profiles/: per-kernel call count / total / average duration
(the --stats view) and per-kernel averages of every collected counter.

    python tools/prof_summary.py OUT.txt DIR_OR_DB [DIR_OR_DB ...]
"""
import glob
import os
import sqlite3
import sys


def dbs(path):
    if os.path.isdir(path):
        return sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))
    return [path]


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    lines = []
    for p in paths:
        for db in dbs(p):
            c = sqlite3.connect(db)
            lines.append("# %s" % os.path.relpath(db))
            try:
                rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
            except sqlite3.Error:
                rows = []
            if rows:
                lines.append("## kernel stats (durations in us)")
                lines.append("calls,total_us,avg_us,pct,kernel")
                for name, calls, total, avg, pct in rows:
                    lines.append("%d,%.3f,%.3f,%.2f,%s" % (calls, total, avg, pct, name[:160]))
            try:      # the same kernel at different grid sizes is a different workload: split the library's kernels by grid
                rows = list(c.execute("select name, grid_x / workgroup_x, count(*), avg(end - start) / 1000.0, min(end - start) / 1000.0 "
                                      "from kernels where name like '%p4s::%' group by 1, 2 order by 1, 2"))
            except sqlite3.Error:
                rows = []
            if rows:
                lines.append("## library kernels by grid (workgroups,calls,avg_us,min_us,kernel)")
                for name, wgs, n, avg, mn in rows:
                    lines.append("%d,%d,%.3f,%.3f,%s" % (wgs, n, avg, mn, name[:120]))
            try:
                rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                                      "from counters_collection group by 1, 2"))
            except sqlite3.Error:
                rows = []
            if rows:
                lines.append("## counters (per dispatch)")
                lines.append("counter,dispatches,avg,min,max,kernel")
                for name, ctr, n, avg, mn, mx in rows:
                    lines.append("%s,%d,%.3f,%.3f,%.3f,%s" % (ctr, n, avg, mn, mx, name[:160]))
            lines.append("")
    with open(out, "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
