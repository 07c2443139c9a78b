#!/usr/bin/env python3
"""The kernels that dominate the LAST run() in a rocprofv3 --kernel-trace CSV of `python3 bench.py ...`, as JSON for bench.py's
roofline.step.top_kernels (reported only when `lib_sha256` matches the library being timed).
usage: python tools/summarize/top_kernels.py <kernel_trace.csv> > profiles/r02_top_kernels.json"""
import collections
import csv
import hashlib
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
ev = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
        wg = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, wg))
ev.sort()
ends = [i for i, e in enumerate(ev) if e[2].startswith("bump_epoch_kernel")]
# bench.py runs the headline program first (warm-up + timed steps): take the last run() BEFORE any other program's kernels
headline = [i for i in ends if i < len(ev)]
first_other = next((i for i, e in enumerate(ev) if e[3][1] >= 4096), len(ev))  # the roofline leg's 4096-limb launches come after the runs
runs = [i for i in ends if i < first_other]
sel = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # which run() (0-based) = warm-up + timed steps of the headline program
lo, hi = (runs[sel - 1] + 1, runs[sel] + 1) if len(runs) > sel and sel > 0 else (0, runs[0] + 1 if runs else len(ev))
run = ev[lo:hi]
wall = run[-1][1] - run[0][0]
busy = sum(e[1] - e[0] for e in run)
by_name, by_grid = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0])
for s, e, n, g in run:
    by_name[n][0] += 1
    by_name[n][1] += e - s
    by_grid[(n, g)][0] += 1
    by_grid[(n, g)][1] += e - s
out = {"source": "rocprofv3 --kernel-trace of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`, one run() of the headline program",
       "lib_sha256": hashlib.sha256((ROOT / "dacapo_amd" / "lib" / "libSEAL_HEVM.so").read_bytes()).hexdigest(),
       "kernels_in_run": len(run), "wall_ms_under_profiler": round(wall / 1e6, 3), "kernel_time_ms": round(busy / 1e6, 3),
       "by_kernel": [{"kernel": n, "calls": c, "avg_us": round(t / c / 1e3, 2), "total_ms": round(t / 1e6, 3), "share_of_kernel_time": round(t / busy, 4)}
                     for n, (c, t) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:8]],
       "by_kernel_and_grid": [{"kernel": n, "grid_workgroups": list(g), "calls": c, "avg_us": round(t / c / 1e3, 2), "total_ms": round(t / 1e6, 3)}
                              for (n, g), (c, t) in sorted(by_grid.items(), key=lambda kv: -kv[1][1])[:8]]}
print(json.dumps(out, indent=1))
