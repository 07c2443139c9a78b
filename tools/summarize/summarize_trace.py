#!/usr/bin/env python3
"""Per-(kernel, grid) average durations from a rocprofv3 --kernel-trace CSV, so that the launches of bench.py's
roofline leg (4096-limb batches) can be read separately from the small launches of the HEVM run in the same command.
usage: python tools/summarize/summarize_trace.py <..._kernel_trace.csv> [min_grid_y]"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(list)
min_y = int(sys.argv[2]) if len(sys.argv) > 2 else 0
trace = list(csv.DictReader(open(sys.argv[1])))
# The single-crossing kernel runs as a persistent grid (one workgroup per CU) whatever the number of limbs, so its grid does not tell the
# leg's 4096-limb launches from the set-up's smaller batches: the leg's are the longest ones (within 20 % of the kernel's maximum).
full_max = max([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace if "ntt_full15_kernel<false" in r["Kernel_Name"]] or [0])
for r in trace:
    if True:
        gy = int(r["Grid_Size_Y"])
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        # the leg's launches: `min_y` limbs on the grid's y axis (the two-launch tiles), or the single-crossing kernel's longest launches
        if gy < min_y and not (min_y and "ntt_full15_kernel<false" in r["Kernel_Name"] and dur >= 0.8 * full_max):
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
        key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), gy, int(r["Grid_Size_Z"]))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':60s} {'grid(x,y,z) in workgroups':>26s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s} {'total_ms':>10s}")
for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{key[0][:60]:60s} {str(key[1:]):>26s} {len(d):7d} {sum(d)/len(d)/1e3:10.2f} {min(d)/1e3:10.2f} {sum(d)/1e6:10.2f}")
