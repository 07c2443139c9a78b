#!/usr/bin/env python3
"""Per-kernel HBM traffic of ONE run() of the headline program, from three rocprofv3 passes of `python3 tools/legs/headline_only.py 3`:
    rocprofv3 --kernel-trace --output-format csv -d kt -- python3 tools/legs/headline_only.py 3
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d pf -- python3 tools/legs/headline_only.py 3
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d pw -- python3 tools/legs/headline_only.py 3
    python tools/summarize/kernel_traffic.py kt/*/*kernel_trace.csv pf/*/*counter_collection.csv pw/*/*counter_collection.csv > profiles/r04_step_kernels.json
Durations come from the counter-free pass; bytes = FETCH_SIZE x 2 (gfx950 counts a 128-byte request of a wide coalesced read as 64:
MI355X_MICROARCH.md) + WRITE_SIZE, both in KB.  For the kernels whose grid encodes (level l, batch B) two byte counts are recomputed
(P_limb = 8 N) and kept APART, so that every fraction can be re-derived from this file:
    section_8d_bytes_of_items   SURVEY.md 8(d)'s figure for the OPS whose items the launch serves -- a key switch at l primes is
                                (2 l^2 + 7 l) P_limb for ALL of its launches together; it contains no intermediate of this design
    design_io_budget_bytes      what THIS launch must read and write in this design: for the fused key-switch middle the l (l + 1)
                                lifted-digit limbs (an intermediate the previous launch wrote), the 2 l (l + 1) key limbs, 2 (l + 1) out.
                                Round 4 called this figure "algorithmic" (and so got traffic / algorithmic < 1): it is a budget, not 8(d)'s bytes
The roofline fraction of a kernel is its MEASURED traffic over its duration (hbm_frac_of_peak)."""
import collections
import csv
import hashlib
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
N, P_LIMB, PEAK = 32768, 8 * 32768, 8.0e12


def short(name):
    return re.sub(r"\(.*", "", name).replace("void dacapo::", "").replace("dacapo::", "")


def last_run(rows, ts_key):
    """the dispatches of the last complete run(): between the last two bump_epoch markers"""
    rows.sort(key=lambda r: int(r[ts_key]))
    ends = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("bump_epoch_kernel")]
    if len(ends) < 2:
        return rows
    return rows[ends[-2] + 1 : ends[-1] + 1]


def grid_wg(r):
    if "Grid_Size_X" in r:
        return (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    return None


kt = last_run(list(csv.DictReader(open(sys.argv[1]))), "Start_Timestamp")
dur = collections.defaultdict(lambda: [0, 0.0])
grids = collections.defaultdict(collections.Counter)
alg = collections.defaultdict(float)  # design_io_budget_bytes
s8d = collections.defaultdict(float)  # section_8d_bytes_of_items
for r in kt:
    n, g = short(r["Kernel_Name"]), grid_wg(r)
    dur[n][0] += 1
    dur[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    grids[n][g] += 1
    # algorithmic bytes of the items a launch processes, where the grid encodes (l, B)
    if n.startswith("f_ks_frows_mac_kernel"):      # grid (tiles, l + 2, B): l (l + 1) operand limbs + 2 l (l + 1) key limbs in, 2 (l + 1) out
        # (the MERGE instantiation, <..., true>, serves both special-prime accumulators from ONE row: grid.y = l + 1.  Round 3's version of
        # this tool took l = grid.y - 2 for it too, priced the merged launches one level too low -- 10 instead of 24 limbs per item at
        # l = 2 -- and reported "traffic / algorithmic = 1.52" for a kernel that moves 0.6-0.7 of its algorithmic bytes from HBM)
        # (round 5: with option ks_items_fast the grid is (tiles, B, rows) when B > 1: the row count is the SMALLER of the two whenever a
        # step has more items than l + 2 rows, and a one-item launch has grid.z = 1 = B either way -- rows <= 15, so take B from the other axis
        # only when it cannot be a row count; steps of <= 15 items are ambiguous and priced as (rows = grid.y) like round 4)
        merged = n.rstrip(">").rstrip().endswith("true")
        rows, B = (g[2], g[1]) if g[1] > 15 >= g[2] and g[2] > 1 else (g[1], g[2])
        l = rows - (1 if merged else 2)
        alg[n] += B * (l + 1) * (3 * l + 2) * P_LIMB
        s8d[n] += B * (2 * l * l + 7 * l) * P_LIMB
    elif n.startswith("b_ks_mac_kernel"):
        l, B = g[1] - 1, g[2]
        alg[n] += B * (l + 1) * (3 * l + 2) * P_LIMB
        s8d[n] += B * (2 * l * l + 7 * l) * P_LIMB
wall = int(kt[-1]["End_Timestamp"]) - int(kt[0]["Start_Timestamp"])
busy = sum(v[1] for v in dur.values())


def pmc(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows = last_run(rows, "Start_Timestamp") if rows and "Start_Timestamp" in rows[0] else rows
    acc = collections.defaultdict(float)
    for r in rows:
        acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    return acc


fetch, write = pmc(sys.argv[2], "FETCH_SIZE"), pmc(sys.argv[3], "WRITE_SIZE")
kernels = []
for n, (calls, t) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:8]:
    hbm = (2.0 * fetch.get(n, 0.0) + write.get(n, 0.0)) * 1024.0
    e = {"kernel": n, "calls": calls, "avg_us": round(t / calls / 1e3, 2), "total_ms": round(t / 1e6, 3), "share_of_kernel_time": round(t / busy, 4),
         "grids_workgroups": [[list(g), c] for g, c in grids[n].most_common(3)],
         "hbm_bytes_in_run": hbm, "hbm_gbs": round(hbm / (t * 1e-9) / 1e9, 1), "hbm_frac_of_peak": round(hbm / (t * 1e-9) / PEAK, 4)}
    if alg.get(n):
        e["design_io_budget_bytes_in_run"] = alg[n]
        e["traffic_over_design_io_budget"] = round(hbm / alg[n], 3)
        e["section_8d_bytes_of_items_in_run"] = s8d[n]
        e["section_8d_note"] = ("8(d)'s bytes of the whole key switches these launches belong to (all of their launches together): this kernel's "
                                "measured traffic is hbm_bytes_in_run, its roofline fraction hbm_frac_of_peak")
    kernels.append(e)
total_hbm = (2.0 * sum(fetch.values()) + sum(write.values())) * 1024.0
out = {"source": "rocprofv3 passes of `python3 tools/legs/headline_only.py 3` (kernel trace; --pmc FETCH_SIZE; --pmc WRITE_SIZE), the last run()",
       "lib_sha256": hashlib.sha256((ROOT / "dacapo_amd" / "lib" / "libSEAL_HEVM.so").read_bytes()).hexdigest(),
       "note": "gfx950: FETCH_SIZE doubled (128-byte requests tallied at 64), WRITE_SIZE exact, KB -> bytes; durations from the counter-free pass",
       "kernels_in_run": len(kt), "wall_ms_under_profiler": round(wall / 1e6, 3), "kernel_time_ms": round(busy / 1e6, 3),
       "bytes_actually_moved_in_run": total_hbm, "bytes_actually_moved_gbs_over_wall": round(total_hbm / (wall * 1e-9) / 1e9, 1),
       "top_kernels": kernels, "dominant": kernels[0] if kernels else None}
print(json.dumps(out, indent=1))
