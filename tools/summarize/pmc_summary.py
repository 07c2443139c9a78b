#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 --pmc counter_collection CSV: python tools/summarize/pmc_summary.py <counter_collection.csv>"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            calls[name] += 1
for name, cs in acc.items():
    print(f"{name}  ({calls[name]} launches)")
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {v / calls[name]:18.1f} per launch")
