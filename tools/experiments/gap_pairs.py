#!/usr/bin/env python3
"""Where the GPU idles inside the last run() of a rocprofv3 --kernel-trace CSV: idle time before a kernel = its start minus the latest
end of everything that started earlier, grouped by (previous kernel -> this kernel).  usage: python tools/experiments/gap_pairs.py <csv> [run_index]"""
import collections
import csv
import re
import sys

ev = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Grid_Size_Y"])))
ev.sort()
ends = [i for i, e in enumerate(ev) if e[2].startswith("bump_epoch_kernel")]
first_other = next((i for i, e in enumerate(ev) if e[3] >= 4096), len(ev))
runs = [i for i in ends if i < first_other]
sel = int(sys.argv[2]) if len(sys.argv) > 2 else len(runs) - 1
lo, hi = (runs[sel - 1] + 1 if sel > 0 else 0), runs[sel] + 1
run = ev[lo:hi]
pairs = collections.defaultdict(lambda: [0, 0])
latest_end, prev, idle_total, overlap = run[0][1], run[0][2], 0, 0
hist = collections.Counter()
for s, e, n, _ in run[1:]:
    gap = s - latest_end
    if gap > 0:
        idle_total += gap
        pairs[(prev, n)][0] += 1
        pairs[(prev, n)][1] += gap
        hist[min(int(gap / 1000), 20)] += 1
    else:
        overlap += 1
    if e > latest_end:
        latest_end, prev = e, n
print(f"run {sel}: {len(run)} kernels, wall {(run[-1][1]-run[0][0])/1e6:.2f} ms, idle {idle_total/1e6:.2f} ms, {overlap} kernels started while another was running")
print("idle-gap histogram (us: count):", dict(sorted(hist.items())))
for (a, b), (c, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6:7.3f} ms  {c:5d} x {t/c/1e3:6.2f} us   {a[:44]:44s} -> {b[:44]}")
