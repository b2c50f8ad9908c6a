#!/usr/bin/env python3
"""HBM bytes per launch of each bench entry from the per-entry rocprofv3 --pmc passes of tools/pmc_traffic.sh.

    python3 tools/traffic_from_pmc.py DIR OUT.json OUT.txt

DIR/<entry>/FETCH_SIZE and DIR/<entry>/WRITE_SIZE hold one rocpd database each.  Every run launches its entry
LAUNCHES times (1 warm-up + 2 timed); the counters of all the library's kernels in the run are summed and divided by
that.  Runs of entries other than the headline still launch the headline workload, shrunk to 16 pages: those
dispatches are dropped.  FETCH_SIZE and WRITE_SIZE are in KiB.  gfx950 tallies a 16-byte-per-lane streaming read at half
its bytes, which is how every kernel here reads its input, so hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE; the raw figures
are kept beside it (the dictionary gather reads single dwords: for it the doubling is an upper bound).
"""
import glob
import json
import os
import sqlite3
import sys

LAUNCHES = 3
SHRUNK_HEADLINE_GRID = 16 * 256
HEADLINE_KEY = "delta_int64_2048x131072"


def per_kernel(db, counter, drop_shrunk):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, grid_size, value from counters_collection where counter_name = ?", (counter,))
    out = {}
    for name, grid, value in rows:
        if "p4s::" not in name:
            continue
        # (16 pages go through the small-batch split: its walk / finish passes, the parts and the redo launch)
        if drop_shrunk and "<unsigned long>" in name and (
                "k_delta_split_" in name or ("k_delta_binary_packed" in name and grid in (SHRUNK_HEADLINE_GRID, 16 * SHRUNK_HEADLINE_GRID))):
            continue
        short = name.split("(")[0].replace("void ", "")
        k = out.setdefault(short, [0, 0.0])
        k[0] += 1
        k[1] += value * 1024.0
    return out


def main():
    root, out_json, out_txt = sys.argv[1:4]
    result, lines = {}, []
    for d in sorted(glob.glob(os.path.join(root, "*", ""))):
        entry = os.path.basename(os.path.dirname(d))
        dbs = {c: glob.glob(os.path.join(d, c, "**", "*_results.db"), recursive=True) for c in ("FETCH_SIZE", "WRITE_SIZE")}
        if not all(dbs.values()):
            continue
        drop = entry != "headline"
        f = per_kernel(dbs["FETCH_SIZE"][0], "FETCH_SIZE", drop)
        w = per_kernel(dbs["WRITE_SIZE"][0], "WRITE_SIZE", drop)
        fetch = sum(v[1] for v in f.values()) / LAUNCHES
        write = sum(v[1] for v in w.values()) / LAUNCHES
        key = HEADLINE_KEY if entry == "headline" else entry
        result[key] = {"fetch_size_bytes": int(fetch), "write_size_bytes": int(write), "hbm_bytes": int(2 * fetch + write),
                       "kernels": {k: {"dispatches_per_launch": round(v[0] / LAUNCHES, 2),
                                       "fetch_size_bytes_per_dispatch": int(v[1] / v[0]),
                                       "write_size_bytes_per_dispatch": int(w.get(k, [1, 0.0])[1] / max(w.get(k, [1])[0], 1))}
                                   for k, v in f.items()}}
        lines.append("%-26s fetch %12d  write %12d  hbm (2 x fetch + write) %12d" % (key, fetch, write, 2 * fetch + write))
        for k, v in f.items():
            lines.append("    %-60s x%-5.2f fetch/dispatch %12d  write/dispatch %12d" % (
                k[:60], v[0] / LAUNCHES, v[1] / v[0], w.get(k, [1, 0.0])[1] / max(w.get(k, [1])[0], 1)))
    result["_comment"] = ("HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes per entry "
                          "(tools/pmc_traffic.sh): hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; gfx950 tallies "
                          "16-byte-per-lane streaming reads at half their bytes")
    with open(out_json, "w") as fh:
        json.dump(result, fh, indent=1, sort_keys=True)
    with open(out_txt, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
