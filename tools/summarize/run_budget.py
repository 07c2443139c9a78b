#!/usr/bin/env python3
"""Kernel-by-kernel budget of ONE run() of a program (the headline or one of its lowerings) from four rocprofv3 passes of
`python3 tools/legs/headline_only.py 3 [b13]` (durations; FETCH_SIZE; WRITE_SIZE; VALU counters):
    python tools/summarize/run_budget.py kt.csv fetch.csv write.csv valu.csv [top=14] [json=out.json] [label=...]
Every kernel NAME of the last complete run (between the last two epoch bumps), all its launches summed:
    measured   total ms, share of kernel time, HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE), VALU wave-instructions (SQ_INSTS_VALU)
    floors     bytes / 5.5 TB/s;  VALU instructions x 4 cycles / 1024 SIMDs / 2.05 GHz  (tools/summarize/per_op_budget.py explains both)
    x floor    measured / max(floors): what the launches of that kernel lose to latency, quantisation and stalls together
A per-launch minimum is not applied here: a program's small launches overlap on the plan's two streams."""
import collections
import csv
import json
import re
import sys

HBM_STREAM, CLOCK, SIMDS = 5.5e12, 2.05e9, 1024


def short(name):
    return re.sub(r"\(.*", "", name).replace("void dacapo::", "").replace("dacapo::", "")


def last_run(rows):
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("bump_epoch_kernel")]
    return rows if len(ends) < 2 else rows[ends[-2] + 1: ends[-1] + 1]


kt, pf, pw, pv = sys.argv[1:5]
opts = dict(a.split("=", 1) for a in sys.argv[5:] if "=" in a)
trace = last_run(list(csv.DictReader(open(kt))))
dur = collections.defaultdict(lambda: [0, 0])
for r in trace:
    n = short(r["Kernel_Name"])
    dur[n][0] += 1
    dur[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
wall = int(trace[-1]["End_Timestamp"]) - int(trace[0]["Start_Timestamp"])
busy = sum(t for _, t in dur.values())


def pmc(path, counters):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] in counters]
    # one row per (dispatch, counter): the last run = the dispatches between the last two epoch bumps
    by_disp = collections.defaultdict(list)
    for r in rows:
        by_disp[int(r["Dispatch_Id"])].append(r)
    order = sorted(by_disp)
    ends = [i for i, d in enumerate(order) if short(by_disp[d][0]["Kernel_Name"]).startswith("bump_epoch_kernel")]
    if len(ends) >= 2:
        order = order[ends[-2] + 1: ends[-1] + 1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in order:
        for r in by_disp[d]:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


fetch, write = pmc(pf, {"FETCH_SIZE"}), pmc(pw, {"WRITE_SIZE"})
valu = pmc(pv, {"SQ_INSTS_VALU"})
top = int(opts.get("top", 14))
rows = []
tot = collections.defaultdict(float)
for n, (calls, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    b = 2.0 * fetch[n].get("FETCH_SIZE", 0.0) * 1024.0 + write[n].get("WRITE_SIZE", 0.0) * 1024.0
    iv = valu[n].get("SQ_INSTS_VALU", 0.0)
    fb, fv = b / HBM_STREAM * 1e3, iv * 4.0 / SIMDS / CLOCK * 1e3  # ms
    fl = max(fb, fv)
    rows.append({"kernel": n, "calls": calls, "total_ms": round(t / 1e6, 3), "share": round(t / busy, 4), "avg_us": round(t / calls / 1e3, 1),
                 "hbm_GB": round(b / 1e9, 3), "hbm_TBps": round(b / (t * 1e-9) / 1e12, 2) if t else 0, "valu_Minst": round(iv / 1e6, 1),
                 "floor_bytes_ms": round(fb, 3), "floor_valu_ms": round(fv, 3), "bound": "valu" if fv > fb else "bytes",
                 "x_floor": round(t / 1e6 / fl, 2) if fl else None})
    tot["ms"] += t / 1e6
    tot["bytes"] += b
    tot["valu"] += iv
    tot["floor"] += fl
out = {"label": opts.get("label", ""), "launches": len(trace), "wall_ms_under_profiler": round(wall / 1e6, 3), "kernel_time_ms": round(busy / 1e6, 3),
       "hbm_GB": round(tot["bytes"] / 1e9, 2), "valu_Ginst": round(tot["valu"] / 1e9, 3), "sum_of_floors_ms": round(tot["floor"], 2),
       "floor_bytes_ms_whole_run": round(tot["bytes"] / HBM_STREAM * 1e3, 2), "floor_valu_ms_whole_run": round(tot["valu"] * 4.0 / SIMDS / CLOCK * 1e3, 2),
       "kernels": rows[:top]}
print(f"== {out['label']}: {out['launches']} launches, wall {out['wall_ms_under_profiler']} ms under the profiler, kernel time {out['kernel_time_ms']} ms; "
      f"{out['hbm_GB']} GB moved (floor {out['floor_bytes_ms_whole_run']} ms at 5.5 TB/s), {out['valu_Ginst']} G VALU wave-instructions "
      f"(floor {out['floor_valu_ms_whole_run']} ms); sum over kernels of max(floors) {out['sum_of_floors_ms']} ms")
print(f"{'kernel':52s} {'calls':>6s} {'ms':>8s} {'share':>6s} {'avg us':>8s} {'GB':>8s} {'TB/s':>5s} {'VALU M':>9s} {'f.bytes':>8s} {'f.valu':>8s} {'bound':>5s} {'x':>5s}")
for r in rows[:top]:
    print(f"{r['kernel'][:52]:52s} {r['calls']:6d} {r['total_ms']:8.2f} {r['share']:6.3f} {r['avg_us']:8.1f} {r['hbm_GB']:8.2f} {r['hbm_TBps']:5.2f} {r['valu_Minst']:9.1f} "
          f"{r['floor_bytes_ms']:8.2f} {r['floor_valu_ms']:8.2f} {r['bound']:>5s} {(r['x_floor'] or 0):5.2f}")
if "json" in opts:
    import hashlib
    from pathlib import Path

    lib = Path(__file__).resolve().parent.parent.parent / "dacapo_amd" / "lib" / "libSEAL_HEVM.so"
    out["lib_sha256"] = hashlib.sha256(lib.read_bytes()).hexdigest() if lib.exists() else None  # bench.py reports a record only for the build it times
    json.dump(out, open(opts["json"], "w"), indent=1)
