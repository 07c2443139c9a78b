#!/usr/bin/env python3
"""VALU occupancy of the single-crossing NTT from one rocprofv3 --pmc pass of `python3 tools/legs/ntt_variant_only.py 1 4096 2`:
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d out -- \\
        python3 tools/legs/ntt_variant_only.py 1 4096 2
    python tools/summarize/ntt_valu.py out/*/*counter_collection.csv out/*/*kernel_trace.csv > profiles/r03_ntt_valu.json
The kernel runs as a persistent grid of one 16-wave workgroup per CU, four waves per SIMD for the whole launch, so a SIMD's VALU is busy for
sum over its waves of SQ_ACTIVE_INST_VALU out of SQ_WAVE_CYCLES / 4 (both count in units of four cycles): that ratio needs no clock.  The
clock figure assumes GRBM_GUI_ACTIVE sums the 8 XCDs."""
import collections
import csv
import hashlib
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
acc, calls, seen = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter(), set()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
        if not name.startswith("ntt_full15_kernel"):
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if (name, r["Dispatch_Id"]) not in seen:
            seen.add((name, r["Dispatch_Id"]))
            calls[name] += 1
dur = collections.defaultdict(list)
if len(sys.argv) > 2:
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
            if name.startswith("ntt_full15_kernel"):
                dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- "
                  "python3 tools/legs/ntt_variant_only.py 1 4096 2",
       "lib_sha256": hashlib.sha256((ROOT / "dacapo_amd" / "lib" / "libSEAL_HEVM.so").read_bytes()).hexdigest(),
       "limbs": 4096, "N": 32768, "kernels": {}}
for name, cs in acc.items():
    n = calls[name]
    c = {k: v / n for k, v in cs.items()}
    waves = 4096 * 16  # one 16-wave pass over every limb
    rec = {"launches": n, "per_launch": c, "valu_instructions_per_wave_per_limb": round(c.get("SQ_INSTS_VALU", 0) / waves, 1)}
    if c.get("SQ_WAVE_CYCLES"):
        rec["simd_valu_busy_frac"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / (c["SQ_WAVE_CYCLES"] / 4.0), 4)
    if dur.get(name):
        us = sum(dur[name]) / len(dur[name])
        rec["avg_us_under_profiler"] = round(us, 1)
        if c.get("GRBM_GUI_ACTIVE"):
            rec["clock_ghz_if_counter_sums_8_xcds"] = round(c["GRBM_GUI_ACTIVE"] / 8.0 / us / 1e3, 3)
    out["kernels"][name] = rec
print(json.dumps(out, indent=1))
