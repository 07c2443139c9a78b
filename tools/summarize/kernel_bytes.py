#!/usr/bin/env python3
"""Per-kernel duration and MEASURED HBM bytes of any command, from three rocprofv3 passes of the same command:
    rocprofv3 --kernel-trace --output-format csv -d kt -- <cmd>      (durations)
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d pf -- <cmd>
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d pw -- <cmd>
    python tools/summarize/kernel_bytes.py kt/*/*kernel_trace.csv pf/*/*counter_collection.csv pw/*/*counter_collection.csv [per=<kernel whose calls count the units>] [top=24]
bytes = FETCH_SIZE x 2 (gfx950 tallies a 128-byte request of a wide coalesced read as 64: MI355X_MICROARCH.md) + WRITE_SIZE, KB -> bytes,
summed over ALL launches of a kernel in the command and divided by the number of units (`per`: e.g. one key switch = one launch of
hyb_mac_kernel<0>).  Prints a table; the last line sums every listed kernel."""
import collections
import csv
import re
import sys


def short(name):
    return re.sub(r"\(.*", "", name).replace("void dacapo::", "").replace("dacapo::", "")


kt, pf, pw = sys.argv[1:4]
opts = dict(a.split("=", 1) for a in sys.argv[4:] if "=" in a)
dur = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(kt)):
    n = short(r["Kernel_Name"])
    dur[n][0] += 1
    dur[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def pmc(path, counter):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    return acc


fetch, write = pmc(pf, "FETCH_SIZE"), pmc(pw, "WRITE_SIZE")
per = opts.get("per")
units = next((c for n, (c, _) in dur.items() if per and n.startswith(per)), 0) or 1
top = int(opts.get("top", 24))
rows = sorted(dur.items(), key=lambda kv: -kv[1][1])[:top]
total_t = sum(t for _, (_, t) in dur.items())
print(f"units: {units} (launches of {per})" if per else "units: 1")
print(f"{'kernel':52s} {'calls':>6s} {'avg us':>9s} {'us/unit':>9s} {'share':>6s} {'read MB/unit':>13s} {'write MB/unit':>13s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
sum_us = sum_r = sum_w = 0.0
for n, (c, t) in rows:
    rd, wr = 2.0 * fetch.get(n, 0.0) * 1024.0, write.get(n, 0.0) * 1024.0
    us_unit = t / 1e3 / units
    gbs = (rd + wr) / (t * 1e-9) / 1e9 if t else 0.0
    sum_us, sum_r, sum_w = sum_us + us_unit, sum_r + rd / units, sum_w + wr / units
    print(f"{n[:52]:52s} {c:6d} {t / c / 1e3:9.1f} {us_unit:9.1f} {t / total_t:6.3f} {rd / units / 1e6:13.1f} {wr / units / 1e6:13.1f} {gbs:8.1f} {gbs / 8000:9.3f}")
print(f"{'sum of the kernels listed':52s} {'':6s} {'':9s} {sum_us:9.1f} {'':6s} {sum_r / 1e6:13.1f} {sum_w / 1e6:13.1f} {(sum_r + sum_w) / (sum_us * 1e-6) / 1e9 if sum_us else 0:8.1f}")
