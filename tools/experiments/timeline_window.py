#!/usr/bin/env python3
"""Consecutive kernels of the last run() in a rocprofv3 --kernel-trace CSV around the n-th launch of a kernel: start offset, duration,
idle time before (single-stream traces: DACAPO_HEVM_PLAN_LANES=1).  usage: timeline_window.py <csv> <kernel substring> [n=20] [width=14]"""
import csv
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 20
width = int(sys.argv[4]) if len(sys.argv) > 4 else 14
ev = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
ends = [i for i, e in enumerate(ev) if e[2].startswith("bump_epoch_kernel")]
run = ev[ends[-2] + 1:ends[-1] + 1] if len(ends) >= 2 else ev
hits = [i for i, e in enumerate(run) if pat in e[2]]
c = hits[min(nth, len(hits) - 1)]
t0 = run[max(0, c - width)][0]
for i in range(max(1, c - width), min(len(run), c + width)):
    s, e, name, gx, gy, gz, wx, qid, sid = run[i]
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:6.2f}  idle before {(s - run[i - 1][1]) / 1e3:6.2f}  {name[:58]:58s} grid {int(gx)//int(wx)}x{gy}x{gz} q{qid} s{sid}")
