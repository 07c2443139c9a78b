#!/usr/bin/env python3
"""Busy/idle split of the GPU timeline from a rocprofv3 --kernel-trace CSV: for the last `--tail-ms` of the trace (the
timed run() calls of bench.py come last, before the roofline leg is excluded by --max-grid-y) print kernel count, summed
kernel time, summed gaps between consecutive kernels, and the per-kernel-name averages of (duration, gap before).
usage: python tools/summarize/timeline_gaps.py <kernel_trace.csv> [--max-grid-y 2000]"""
import argparse
import collections
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--max-grid-y", type=int, default=2000)
a = ap.parse_args()
ev = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Grid_Size_Y"])))
ev.sort()
# find the runs: bump_epoch_kernel ends every run()
ends = [i for i, e in enumerate(ev) if e[2].startswith("bump_epoch_kernel")]
if len(ends) >= 2:
    lo, hi = ends[-2] + 1, ends[-1] + 1
else:
    lo, hi = 0, len(ev)
run = [e for e in ev[lo:hi] if e[3] <= a.max_grid_y]
busy = sum(e[1] - e[0] for e in run)
gaps = [max(0, run[i][0] - run[i - 1][1]) for i in range(1, len(run))]
wall = run[-1][1] - run[0][0]
print(f"last run(): {len(run)} kernels, wall {wall/1e6:.2f} ms, kernel time {busy/1e6:.2f} ms, gaps {sum(gaps)/1e6:.2f} ms "
      f"(avg kernel {busy/len(run)/1e3:.2f} us, avg gap {sum(gaps)/max(1,len(gaps))/1e3:.2f} us)")
per = collections.defaultdict(lambda: [0, 0, 0])
for i, e in enumerate(run):
    p = per[e[2]]
    p[0] += 1
    p[1] += e[1] - e[0]
    p[2] += gaps[i - 1] if i else 0
print(f"{'kernel':52s} {'calls':>6s} {'avg_us':>8s} {'gap_before_us':>14s} {'total_ms':>9s}")
for k, p in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:30]:
    print(f"{k[:52]:52s} {p[0]:6d} {p[1]/p[0]/1e3:8.2f} {p[2]/p[0]/1e3:14.2f} {(p[1]+p[2])/1e6:9.2f}")
