#!/usr/bin/env python3
"""Kernel-by-kernel budget of ONE op (a rotation hop, ct x ct + relinearise, a rescale at 13 primes; config 3) from four rocprofv3 passes of
`python3 tools/legs/per_op_only.py <iters> --only <op>` (durations; FETCH_SIZE; WRITE_SIZE; VALU counters):
    python tools/summarize/per_op_budget.py <op> <iters> kt.csv fetch.csv write.csv valu.csv [json=out.json]
Per launch of every kernel of the op, in launch order:
    measured   avg us (kernel trace), HBM bytes = FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, VALU wave-instructions
    floors     bytes / 5.5 TB/s (what a streaming kernel reaches on this part: the copy kernel's 4.7-5.6 TB/s, the n-ary sums' 5.6)
               VALU: SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.05 GHz (a wave64 instruction holds its SIMD for four cycles; the clock under load)
               the same x ceil(workgroups / 256) / (workgroups / 256): equal workgroups dealt to 256 CUs -- the busiest CU's share over the mean
               (480 workgroups: 2 against 1.875)
               3.7 us per launch: what a minimal dependent launch costs in a chain (a hop at 1 prime = 5 launches = 18.4 us,
               profiles/r02_chain_latency.txt) -- a kernel cannot be shorter than its own workgroup's load -> phase -> store latency + the gap
    floor      max of these; `x floor` = measured / floor
The op's own line sums the launches of one iteration: measured kernel time, the sum of the floors, and the HIP-event time of the unprofiled loop."""
import collections
import csv
import json
import math
import re
import sys

HBM_STREAM = 5.5e12
CLOCK = 2.05e9
SIMDS = 1024
CUS = 256
LAUNCH_MIN_US = 3.7


def short(name):
    return re.sub(r"\(.*", "", name).replace("void dacapo::", "").replace("dacapo::", "")


op, iters = sys.argv[1], int(sys.argv[2])
kt, pf, pw, pv = sys.argv[3:7]
opts = dict(a.split("=", 1) for a in sys.argv[7:] if "=" in a)

Launch = collections.namedtuple("Launch", "name wgs wg_size vgpr lds t0 dur")
trace = []
for r in csv.DictReader(open(kt)):
    n = short(r["Kernel_Name"])
    if n.startswith("__amd_rocclr"):
        continue
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    wgs = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // wg
    # (rocprofv3's VGPR_Count is half the code object's .vgpr_count on gfx950 -- 76 for a kernel compiled to 152: reported x 2)
    trace.append(Launch(n, wgs, wg, 2 * (int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)), int(r.get("LDS_Block_Size", 0) or 0),
                        int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
trace.sort(key=lambda l: l.t0)
# the timed loop = the LAST iters repetitions of the op's launch sequence; the sequence = the distinct (kernel, grid) pairs of the trace's tail
# whose count is a multiple of iters + 1 (one warm-up call precedes the loop)
by_key = collections.defaultdict(list)
for l in trace:
    by_key[(l.name, l.wgs)].append(l)
excl = tuple(filter(None, opts.get("exclude", "").split(",")))  # (config 3's process also runs the grouped-digit variant: exclude=hyb)
seq = [(k, v) for k, v in by_key.items() if len(v) % (iters + 1) == 0 and not (excl and k[0].startswith(excl))]
seq.sort(key=lambda kv: kv[1][-1].t0)  # order of the last iteration
per_iter = {k: len(v) // (iters + 1) for k, v in seq}


def pmc(path, counters):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in counters:
            wg = int(r["Workgroup_Size"]) if "Workgroup_Size" in r and r["Workgroup_Size"] else 0
            gs = int(r["Grid_Size"]) if "Grid_Size" in r and r["Grid_Size"] else 0
            key = (short(r["Kernel_Name"]), gs // wg if wg else 0)
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[key].add(r["Dispatch_Id"])
    return {k: {c: v / max(1, len(cnt[k])) for c, v in cs.items()} for k, cs in acc.items()}


fetch, write = pmc(pf, {"FETCH_SIZE"}), pmc(pw, {"WRITE_SIZE"})
valu = pmc(pv, {"SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})


def lookup(tab, key):
    if key in tab:
        return tab[key]
    cands = [v for k, v in tab.items() if k[0] == key[0]]  # (a counter file without grid columns: by name)
    return cands[0] if len(cands) == 1 else {}


rows, tot_us, tot_floor = [], 0.0, 0.0
for key, ls in seq:
    name, wgs = key
    l0 = ls[-1]
    us = sum(l.dur for l in ls[per_iter[key]:]) / (len(ls) - per_iter[key]) / 1e3  # (without the warm-up call)
    rd = 2.0 * lookup(fetch, key).get("FETCH_SIZE", 0.0) * 1024.0
    wr = lookup(write, key).get("WRITE_SIZE", 0.0) * 1024.0
    v = lookup(valu, key)
    insts = v.get("SQ_INSTS_VALU", 0.0)
    waves_per_wg = max(1, l0.wg_size // 64)
    # workgroups a CU holds: 512 VGPRs per SIMD lane, waves of a workgroup spread over 4 SIMDs; LDS 160 KiB
    waves_per_simd = max(1, 512 // max(1, l0.vgpr)) if l0.vgpr else 8
    wg_per_cu = max(1, min(8 * 4 // waves_per_wg, waves_per_simd * 4 // waves_per_wg, (160 * 1024) // max(1, l0.lds) if l0.lds else 64))
    slots = CUS * wg_per_cu
    rounds = wgs / slots
    per_cu = wgs / CUS
    quant = math.ceil(per_cu) / per_cu if per_cu > 0 else 1.0
    f_bytes = (rd + wr) / HBM_STREAM * 1e6
    f_valu = insts * 4.0 / SIMDS / CLOCK * 1e6
    f_valu_q = f_valu * quant
    floor = max(f_bytes, f_valu_q, LAUNCH_MIN_US)
    n = per_iter[key]
    rows.append({"kernel": name, "launches_per_op": n, "workgroups": wgs, "threads": l0.wg_size, "vgprs": l0.vgpr, "lds_bytes": l0.lds,
                 "workgroups_per_cu": wg_per_cu, "rounds_of_the_chip": round(rounds, 3), "avg_us": round(us, 2),
                 "read_MB": round(rd / 1e6, 2), "write_MB": round(wr / 1e6, 2), "hbm_TBps": round((rd + wr) / (us * 1e-6) / 1e12, 3) if us else 0,
                 "valu_wave_instructions": round(insts), "valu_busy_frac": round(v.get("SQ_ACTIVE_INST_VALU", 0) / (v["SQ_WAVE_CYCLES"] / 4.0), 3) if v.get("SQ_WAVE_CYCLES") else None,
                 "floor_bytes_us": round(f_bytes, 2), "floor_valu_us": round(f_valu, 2), "floor_valu_quantised_us": round(f_valu_q, 2),
                 "floor_us": round(floor, 2), "x_floor": round(us / floor, 2) if floor else None})
    tot_us += us * n
    tot_floor += floor * n

out = {"op": op, "iterations": iters, "kernels": rows, "kernel_time_per_op_us": round(tot_us, 1), "sum_of_floors_us": round(tot_floor, 1),
       "assumptions": {"hbm_streaming_TBps": HBM_STREAM / 1e12, "clock_GHz": CLOCK / 1e9, "simds": SIMDS, "valu_cycles_per_wave_instruction": 4,
                       "min_us_per_dependent_launch": LAUNCH_MIN_US}}
if "event_us" in opts:
    out["hip_event_us_unprofiled"] = float(opts["event_us"])
print(f"== {op}: {len(rows)} kernels per op, kernel time {tot_us:.1f} us, sum of floors {tot_floor:.1f} us" + (f", HIP events (unprofiled loop) {opts['event_us']} us" if "event_us" in opts else ""))
print(f"{'kernel':50s} {'n':>2s} {'wgs':>6s} {'vgpr':>4s} {'wg/cu':>5s} {'rounds':>6s} {'avg us':>7s} {'rd MB':>7s} {'wr MB':>7s} {'TB/s':>5s} {'VALU Minst':>10s} {'busy':>5s} {'f.bytes':>7s} {'f.valu':>6s} {'f.v.q':>6s} {'floor':>6s} {'x':>5s}")
for r in rows:
    print(f"{r['kernel'][:50]:50s} {r['launches_per_op']:2d} {r['workgroups']:6d} {r['vgprs']:4d} {r['workgroups_per_cu']:5d} {r['rounds_of_the_chip']:6.2f} {r['avg_us']:7.1f} "
          f"{r['read_MB']:7.1f} {r['write_MB']:7.1f} {r['hbm_TBps']:5.2f} {r['valu_wave_instructions'] / 1e6:10.3f} {(r['valu_busy_frac'] or 0):5.2f} "
          f"{r['floor_bytes_us']:7.1f} {r['floor_valu_us']:6.1f} {r['floor_valu_quantised_us']:6.1f} {r['floor_us']:6.1f} {(r['x_floor'] or 0):5.2f}")
if "json" in opts:
    import hashlib
    from pathlib import Path

    lib = Path(__file__).resolve().parent.parent.parent / "dacapo_amd" / "lib" / "libSEAL_HEVM.so"
    out["lib_sha256"] = hashlib.sha256(lib.read_bytes()).hexdigest() if lib.exists() else None  # bench.py reports a record only for the build it times
    json.dump(out, open(opts["json"], "w"), indent=1)
