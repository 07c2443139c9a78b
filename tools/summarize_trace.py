#!/usr/bin/env python3
"""Per-(kernel, grid) average durations from a rocprofv3 --kernel-trace CSV, so that the launches of bench.py's
roofline leg (4096-limb batches) can be read separately from the small launches of the HEVM run in the same command.
usage: python tools/summarize_trace.py <..._kernel_trace.csv> [min_grid_y]"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(list)
min_y = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        gy = int(r["Grid_Size_Y"])
        gx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        # the leg's launches: `min_y` limbs on the grid's y axis (the two-launch tiles) or as `min_y` workgroups (the single-crossing kernel)
        if gy < min_y and not (min_y and gx >= min_y and "ntt_full" in r["Kernel_Name"]):
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
        key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), gy, int(r["Grid_Size_Z"]))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':60s} {'grid(x,y,z) in workgroups':>26s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s} {'total_ms':>10s}")
for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{key[0][:60]:60s} {str(key[1:]):>26s} {len(d):7d} {sum(d)/len(d)/1e3:10.2f} {min(d)/1e3:10.2f} {sum(d)/1e6:10.2f}")
