#!/usr/bin/env python3
"""Writes profiles/README.md FROM the files it describes: every number in the round-3 / round-4 tables is read out of the JSON / CSV / text
file on the same row, so prose and evidence cannot drift apart (the round-3 review found the hand-written table stale against its files).
    python tools/profiles_readme.py > profiles/README.md
The tables of rounds 1 and 2 are history and stay as written (profiles/README.history.md)."""
import csv
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "profiles"


def J(name):
    f = P / name
    return json.loads(f.read_text()) if f.exists() else None


def short(n):
    return re.sub(r"\(.*", "", n).replace("void dacapo::", "").replace("dacapo::", "")


def stats(name, top=4):
    f = P / name
    if not f.exists():
        return "(missing)"
    rows = list(csv.DictReader(open(f)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    return "; ".join(f"`{short(r['Name'])}` {100 * int(r['TotalDurationNs']) / tot:.1f} % ({int(r['Calls'])} × {float(r['AverageNs']) / 1e3:.0f} µs)" for r in rows[:top]) + \
        f"; {tot / 1e9:.2f} s of kernel time in the file"


def bench_row(r):
    d = J(f"{r}_bench.json")
    if not d:
        return f"| `{r}_bench.json` | (missing) | `python bench.py` |"
    roof, c4, per = d["roofline"], d.get("config4_resnet20_nt65536_N131072") or {}, d["per_op_13_primes"]
    traffic = roof.get("traffic")
    leg = roof["launch"]["avg_us"]
    txt = (f"the un-profiled `python bench.py` line: headline {d['ms_per_step']:.1f} ms / {d['value'] / 1e6:.2f} M NTT/s, roofline leg {leg:.0f} µs = {roof['frac']:.3f} "
           f"(two-launch transform on the same buffer: {roof['two_launch_transform']['avg_us']:.0f} µs)"
           + (f", `roofline.traffic` {traffic / 1e9:.2f} GB = {traffic / roof['launch']['algorithmic_bytes']:.2f}× algorithmic" if traffic else ", `roofline.traffic` null (counter file of another build)")
           + f", config 4 {c4.get('run_s')} s / rms {c4.get('rms_vs_torch', 0):.1e}, cfg3 {d['cfg3_mul_relin']['us']:.0f} µs ({d['cfg3_mul_relin']['grouped_digit_keys']['us']:.0f} µs under grouped-digit keys), "
           f"per op at 13 primes {per['rotate_hop']['us']:.0f} / {per['mulcc_relin']['us']:.0f} / {per['rescale']['us']:.0f} µs, CPU baseline {d['cpu_baseline']['value']:.0f} NTT/s on 1 thread")
    if c4.get("chains"):
        m = c4["chains"]["chain_mixed"]
        txt += f"; config 4 on the mixed chain {m.get('run_s')} s (log2 QP {m.get('log2_QP')}, rms {m.get('rms_vs_torch', 0):.1e})"
    if c4.get("key_sets"):
        ks = c4["key_sets"]
        txt += "; key sets " + ", ".join(f"{k}: {v.get('run_s')} s / {v.get('rotation_key_bytes', 0) / 1e9:.0f} GB / {v.get('key_switches')} switches" for k, v in ks.items() if isinstance(v, dict))
    return f"| `{r}_bench.json` | {txt} | `python bench.py` |"


def bench_row6(r):
    d = J(f"{r}_bench_full.json")
    if not d:
        return f"| `{r}_bench_full.json` | (missing) | `python bench.py --full` |"
    roof, c4, per = d["roofline"], d.get("config4_resnet20_nt65536_N131072") or {}, d["per_op_13_primes"]
    traffic = roof.get("traffic")
    line = (P / f"{r}_bench_default_line.json").read_text().strip().splitlines()[-1] if (P / f"{r}_bench_default_line.json").exists() else ""
    secs = (P / f"{r}_bench_default_seconds.txt").read_text().strip() if (P / f"{r}_bench_default_seconds.txt").exists() else ""
    lz, dh = c4.get("lazy_sums") or {}, c4.get("lazy_sums_double_hoist") or {}
    txt = (f"the full record of `python bench.py --full`: headline {d['ms_per_step']:.1f} ms / {d['value'] / 1e6:.2f} M NTT/s, roofline leg {roof['launch']['avg_us']:.0f} µs = {roof['frac']:.3f} "
           f"(two-launch transform on the same buffer: {roof['two_launch_transform']['avg_us']:.0f} µs)"
           + (f", `roofline.traffic` {traffic / 1e9:.2f} GB = {traffic / roof['launch']['algorithmic_bytes']:.2f}× algorithmic" if traffic else ", `roofline.traffic` null (counter file of another build)")
           + f", config 4 with default options {c4.get('run_s')} s / rms {c4.get('rms_vs_torch', 0):.1e}, with lazy sums {lz.get('run_s')} s, with lazy sums + double hoisting {dh.get('run_s')} s"
           f" ({(dh.get('lazy_sums') or {}).get('rotations')} rotations in {(dh.get('lazy_sums') or {}).get('groups')} groups), cfg3 {d['cfg3_mul_relin']['us']:.0f} µs"
           + (f" ({d['cfg3_mul_relin']['grouped_digit_keys']['us']:.0f} µs under grouped-digit keys)" if d['cfg3_mul_relin'].get('grouped_digit_keys') else "")
           + f", per op at 13 primes {per['rotate_hop']['us']:.0f} / {per['mulcc_relin']['us']:.0f} / {per['rescale']['us']:.0f} µs, CPU baseline {d['cpu_baseline']['value']:.0f} NTT/s on 1 thread")
    kt = d.get("ks_traffic") or {}
    if kt:
        txt += "; measured HBM bytes ÷ SURVEY 8(d) bytes: " + ", ".join(f"{k} {v['traffic_over_algorithmic']}" for k, v in kt.items())
    if c4.get("key_sets"):
        ks = c4["key_sets"]
        txt += "; key sets (default options) " + ", ".join(f"{k}: {v.get('run_s')} s / {v.get('rotation_key_bytes', 0) / 1e9:.0f} GB" for k, v in ks.items() if isinstance(v, dict))
    row = f"| `{r}_bench_full.json` | {txt} | `python bench.py --full --out …` |"
    if line:
        row += (f"\n| `{r}_bench_default_line.json`, `{r}_bench_default_full.json` | the DEFAULT run as the driver starts it (`--gpus 1 --steps 20 --warmup 5`): last stdout line = {len(line)} bytes "
                f"(limit 4 096), {secs}; `value` {json.loads(line)['value'] / 1e6:.3f} M NTT/s, `ms_per_step` {json.loads(line)['ms_per_step']} | `python bench.py --gpus 1 --steps 20 --warmup 5` |")
    return row


def traffic_row(r):
    d = J(f"{r}_ntt_hbm_traffic.json")
    if not d:
        return None
    b, a = d["forward_ntt_hbm_bytes"], d["forward_ntt_algorithmic_bytes"]
    return (f"| `{r}_ntt_hbm_traffic.json` | FETCH_SIZE (×2) + WRITE_SIZE of the forward transform as `dc_ntt_forward` launches it on {d['limbs']} limbs "
            f"({', '.join('`' + k + '`' for k in d['forward_ntt_kernels'])}): {b / 1e9:.3f} GB = {b / a:.2f} × {a / 1e9:.3f} GB; library sha256 `{d['lib_sha256'][:12]}…` | "
            "`rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/legs/ntt_only.py 15 4096 2` (and `WRITE_SIZE`), `tools/summarize/collect_traffic.py` |")


def step_row(r):
    d = J(f"{r}_step_kernels.json")
    if not d:
        return None
    k = d["dominant"]
    if k.get("design_io_budget_bytes_in_run"):  # round 5's fields: 8(d)'s bytes of the ops and the launch's own I/O budget kept apart
        extra = (f" = {k['hbm_frac_of_peak']} of 8 TB/s; this design's I/O budget for those launches {k['design_io_budget_bytes_in_run'] / 1e9:.1f} GB (traffic ÷ budget "
                 f"{k['traffic_over_design_io_budget']}), SURVEY 8(d) bytes of the key switches they belong to {k['section_8d_bytes_of_items_in_run'] / 1e9:.1f} GB")
    else:
        extra = f", algorithmic {k['algorithmic_bytes_in_run'] / 1e9:.1f} GB, traffic ÷ algorithmic {k['traffic_over_algorithmic']}" if k.get("algorithmic_bytes_in_run") else ""
    return (f"| `{r}_step_kernels.json` | the timed step's own kernels: {d['kernels_in_run']} launches in the last `run()`, {d['kernel_time_ms']:.1f} ms of kernel time in "
            f"{d['wall_ms_under_profiler']:.1f} ms under the profiler, {d['bytes_actually_moved_in_run'] / 1e9:.1f} GB actually moved; dominant `{k['kernel']}` "
            f"{k['calls']} × {k['avg_us']} µs, {k['hbm_bytes_in_run'] / 1e9:.2f} GB from HBM{extra} | three passes of `python3 tools/legs/headline_only.py 3`, `tools/summarize/kernel_traffic.py` |")


def valu_row(r):
    d = J(f"{r}_ntt_valu.json")
    if not d:
        return None
    k = next((v for n, v in d.get("kernels", {}).items() if n.startswith("ntt_full15_kernel<false")), {})
    ki = next((v for n, v in d.get("kernels", {}).items() if n.startswith("ntt_full15_kernel<true")), {})
    inv = (f"; inverse kernel: {ki.get('valu_instructions_per_wave_per_limb')} instructions, busy {ki.get('simd_valu_busy_frac')}, {ki.get('avg_us_under_profiler')} µs under the counters"
           if ki else "")
    return (f"| `{r}_ntt_valu.json` | VALU occupancy of the single-crossing forward kernel: {k.get('valu_instructions_per_wave_per_limb')} vector instructions per wave per limb, "
            f"SIMD vector ALUs busy {k.get('simd_valu_busy_frac')} of the launch{inv} | `tools/collect_profiles.sh` B4b, `tools/summarize/ntt_valu.py` |")


def budget_head(name):
    """the summary lines (== op / run: ...) of a tools/summarize/per_op_budget.py or run_budget.py table"""
    f = P / name
    if not f.exists():
        return "(missing)"
    return " / ".join(ln[3:].strip() for ln in f.read_text().splitlines() if ln.startswith("== "))


def budget_rows(name, pat, n=3):
    """rows of a tools/summarize/run_budget.py table as prose: kernel, calls, ms, share, avg us, GB, TB/s, VALU M, floors, bound, x"""
    f = P / name
    if not f.exists():
        return "(missing)"
    out = []
    for ln in f.read_text().splitlines():
        m = re.match(r"\s*(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(valu|bytes)\s+([\d.]+)\s*$", ln)
        if m and re.search(pat, m.group(1)):
            k, calls, ms, share, avg, gb, tbs, vm, fb, fv, bound, x = m.groups()
            out.append(f"`{k}` {calls} × {float(avg):.0f} µs = {float(ms):.1f} ms ({100 * float(share):.1f} %), {float(gb):.0f} GB at {tbs} TB/s, {float(vm) / 1e3:.1f} G VALU instructions: "
                       f"floors {float(fb):.1f} ms (bytes) / {float(fv):.1f} ms (VALU), {float(x):.2f}× its {bound} floor")
        if len(out) >= n:
            break
    return "; ".join(out) if out else "(no such row)"


def sweep_lines(name, n=6):
    f = P / name
    if not f.exists():
        return "(missing)"
    return " // ".join(re.sub(r"\s+", " ", ln.strip()) for ln in f.read_text().splitlines()[:n])


def check_lines(name):
    f = P / name
    if not f.exists():
        return "(missing)"
    out = []
    for ln in f.read_text().splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            out.append(f"{d['limbs']}: fwd {d['fwd_full_us']:.0f} / {d['fwd_two_phase_us']:.0f}, inv {d['inv_full_us']:.0f} / {d['inv_two_phase_us']:.0f}")
    return "; ".join(out)


def first_lines(name, pat, n=3):
    f = P / name
    if not f.exists():
        return "(missing)"
    hits = [ln.strip() for ln in f.read_text().splitlines() if re.search(pat, ln)]
    return " / ".join(hits[:n])


def kb_rows(name, pat, n=3):
    """rows of a tools/summarize/kernel_bytes.py table (kernel, calls, avg us, us/unit, share, read MB/unit, write MB/unit, GB/s, fraction of 8 TB/s) as prose"""
    f = P / name
    if not f.exists():
        return "(missing)"
    out = []
    for ln in f.read_text().splitlines():
        if not re.search(pat, ln):
            continue
        m = re.match(r"\s*(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m:
            k, calls, avg, _, share, rd, wr, gbs, frac = m.groups()
            out.append(f"`{k}` {calls} × {float(avg):.0f} µs, {100 * float(share):.1f} % of the process's kernel time, {float(rd) / 1e3:.1f} GB read + {float(wr) / 1e3:.1f} GB written = {float(gbs) / 1e3:.2f} TB/s ({float(frac):.2f} of peak)")
        if len(out) >= n:
            break
    return "; ".join(out) if out else "(no such row)"


def seq_totals(name):
    f = P / name
    if not f.exists():
        return "(missing)"
    out = []
    for ln in f.read_text().splitlines():
        m = re.match(r"sum of the kernels listed\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", ln)
        if m:
            us, rd, wr, gbs = (float(x) for x in m.groups())
            out.append(f"{us:.0f} µs of kernels, {rd / 1e3:.2f} GB read + {wr / 1e3:.2f} GB written = {(rd + wr) / 1e3:.2f} GB")
    return "; ".join(f"`hyb_fuse` = {f}: {v}" for f, v in zip((2, 1, 0), out))


def hop_levels(name, labels=("`hyb_fuse` = 2", "`hyb_fuse` = 1", "`hyb_fuse` = 0")):
    f = P / name
    if not f.exists():
        return "(missing)"
    out = []
    for ln in f.read_text().splitlines():
        if ln.startswith('{"N"'):
            d = json.loads(ln)
            out.append(f"({d['special']}, {d['alpha']}) levels " + " / ".join(str(l["level"]) for l in d["levels"]) + ": " + " / ".join(f"{l['hop_us']:.0f}" for l in d["levels"]))
    return "; ".join(f"{f}: {v} µs" for f, v in zip(labels, out))


print("# profiles/ — rocprofv3 evidence (MI355X, ROCm 7.2)\n")
print("Generated by `python tools/profiles_readme.py > profiles/README.md`: every figure in the round-3 ... round-6 tables is read from the file on its row.\n")
print("## Round 6 (everything `r06_*`; one `gpurun` call of `tools/collect_profiles.sh r06` on the committed build -- the JSON files that `bench.py` reads carry "
      "the library's sha256)\n")
print("| file | what (figures read from the file) | command |\n|---|---|---|")
rows = [bench_row6("r06"), traffic_row("r06"), step_row("r06"), valu_row("r06"),
        f"| `r06_per_op_kernel_bytes.txt`, `r06_per_op_budget_*.json` | the single ops at 13 primes and config 3 kernel by kernel (duration, measured HBM bytes, VALU wave-instructions, three floors per launch): "
        f"{budget_head('r06_per_op_kernel_bytes.txt')} | `tools/collect_per_op_budget.sh r06` |",
        f"| `r06_run_budget_headline.txt` / `.json`, `r06_run_budget_b13.txt` / `.json` | one `run()` kernel by kernel with the same floors: {budget_head('r06_run_budget_headline.txt')} /// {budget_head('r06_run_budget_b13.txt')} | `tools/collect_run_budget.sh r06 headline` / `b13` |",
        "| `r06_cluster_barrier.txt` | XCD-local / cross-XCD / device-wide barrier costs beside a dependent launch chain, with what they mean for fusing the headline's launch-floor-sized chains (negative) | `tools/experiments/cluster_barrier_bench.hip` |",
        "| `r06_headline_step_profile.txt` | the headline plan step by step (kind, level, batch bucket: steps, total and mean time with a synchronise after every step), the plan's edges by kind, the dataflow graph's width | `DACAPO_HEVM_OPTIONS=plan_graph=0,step_profile=1,trace=1 python tools/legs/headline_only.py 2` |",
        "| `r06_experiments.txt` | what was measured on the way: the barriers, opcode 10's inverse phase folded into the re-encoding kernel (no gain), table indirection (no cost), double hoisting at config 4 | — |",
        f"| `r06_config4_kernel_stats.csv`, `r06_config4_under_profiler.txt` | BASELINE config 4 (default options) under the kernel trace: {stats('r06_config4_kernel_stats.csv')} | `rocprofv3 --kernel-trace --stats -- python3 tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8` |",
        f"| `r06_ntt_full_check.txt` | single-crossing kernel / two-launch tiles, µs, by limb count: {check_lines('r06_ntt_full_check.txt')} | `tools/legs/ntt_full_check.py <limbs> 20` |",
        f"| `r06_kernel_stats.csv`, `r06_by_kernel_and_grid.txt`, `r06_timeline.txt`, `r06_top_kernels.json`, `r06_roofline_leg_launches.txt` | the bench command under the kernel trace: {first_lines('r06_timeline.txt', 'last run', 1)} | `rocprofv3 --kernel-trace --stats … -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config4` |",
        "| `r06_hybrid_ks_kernels.txt`, `r06_boot_kernel_bytes.txt`, `r06_per_op.json`, `r06_per_op_kernel_stats.csv`, `r06_lowering_sweep.txt`, `r06_per_op_sweep.txt`, `r06_chain_latency.txt`, `r06_profiled_SEAL_MI355X.json` | as in round 5 on this build | see round 5's rows |"]
print("\n".join(r for r in rows if r))
print()
print("## Round 5 (everything `r05_*`; one `gpurun` call of `tools/collect_profiles.sh r05` on the committed build -- the JSON files that `bench.py` reads carry "
      "the library's sha256)\n")
print("| file | what (figures read from the file) | command |\n|---|---|---|")
rows = [bench_row("r05"), traffic_row("r05"), step_row("r05"), valu_row("r05"),
        f"| `r05_per_op_kernel_bytes.txt`, `r05_per_op_budget_*.json` | the single ops at 13 primes and config 3 KERNEL BY KERNEL: duration, measured HBM bytes (FETCH_SIZE × 2 + WRITE_SIZE), "
        f"VALU wave-instructions, and three floors per launch (bytes ÷ 5.5 TB/s; VALU × 4 cycles ÷ 1024 SIMDs ÷ 2.05 GHz × CU quantisation; 3.7 µs per dependent launch): "
        f"{budget_head('r05_per_op_kernel_bytes.txt')} | `tools/collect_per_op_budget.sh r05`: four rocprofv3 passes per op, one op per process, `tools/summarize/per_op_budget.py` |",
        f"| `r05_run_budget_b13.txt` / `.json` | one `run()` of the 13-prime lowering kernel by kernel with the same floors: {budget_head('r05_run_budget_b13.txt')}; "
        f"{budget_rows('r05_run_budget_b13.txt', 'f_ks_frows_mac_kernel<8, 2, 0, true>|f_ks_lift_fcols_kernel<7, 3>', 2)} | `tools/collect_run_budget.sh r05 b13`, `tools/summarize/run_budget.py` |",
        f"| `r05_run_budget_headline.txt` / `.json` | the same for the headline program: {budget_head('r05_run_budget_headline.txt')}; "
        f"{budget_rows('r05_run_budget_headline.txt', 'f_ks_frows_mac_kernel<8, 2, 0, true>|f_dr_icols_lift_fcols_kernel<7, 1, false>', 2)} | `tools/collect_run_budget.sh r05 headline` |",
        f"| `r05_lowering_sweep.txt`, `r05_per_op_sweep.txt` | this round's launch-shape options switched off one at a time on one box (HIP events / best of 6 runs): {sweep_lines('r05_lowering_sweep.txt', 5)} //// {sweep_lines('r05_per_op_sweep.txt', 5)} | `tools/legs/lowering_sweep.py`, `tools/legs/per_op_sweep.py` |",
        f"| `r05_ntt_full_check.txt` | single-crossing kernel / two-launch tiles, µs, by limb count (bit-exactness checked in the same run): {check_lines('r05_ntt_full_check.txt')} | `tools/legs/ntt_full_check.py <limbs> 20` |",
        f"| `r05_hybrid_ks_kernels.txt` | one grouped-digit rotation hop at N = 2^17 under round 5's key shape (4 digits of 8 under 9 special primes), kernel by kernel with measured HBM bytes for the three "
        f"launch sequences, the matrix-core counters, all levels under HIP events: {hop_levels('r05_hybrid_ks_kernels.txt', ('rounds 3-4 shape, `hyb_fuse` = 2', '`hyb_fuse` = 2', '`hyb_fuse` = 1', '`hyb_fuse` = 0'))}. "
        f"Per hop at level 31 under the trace: {seq_totals('r05_hybrid_ks_kernels.txt')} | `tools/collect_profiles.sh` B4: `tools/legs/hybrid_ks_bench.py 17 40 9 8 10 31 --opt hyb_fuse=f`, `tools/summarize/kernel_bytes.py` |",
        f"| `r05_config4_kernel_stats.csv`, `r05_config4_under_profiler.txt` | BASELINE config 4 under the kernel trace: {stats('r05_config4_kernel_stats.csv')} | "
        "`rocprofv3 --kernel-trace --stats -- python3 tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8` |",
        f"| `r05_boot_kernel_bytes.txt` | ONE real bootstrap at config 4's geometry, per kernel the measured HBM bytes and GB/s: {first_lines('r05_boot_kernel_bytes.txt', 'bootstrap:', 1)}; "
        f"{kb_rows('r05_boot_kernel_bytes.txt', 'hyb_mac_kernel<0>|hyb_conv_mfma_kernel<true, true, 2>|hyb_conv_mfma_kernel<false, true, 1>|ntt_phase_kernel<8, 3, true, false, false>', 4)} | three passes of `tools/legs/boot_demo.py 17 5 1 14 9 8`, `tools/summarize/kernel_bytes.py` |",
        f"| `r05_per_op.json`, `r05_per_op_kernel_stats.csv` | the three expensive opcodes at 13 primes and config 3 under `--stats`: {stats('r05_per_op_kernel_stats.csv', 3)} | `rocprofv3 --kernel-trace --stats -- python3 tools/legs/per_op_only.py 20` |",
        f"| `r05_kernel_stats.csv`, `r05_by_kernel_and_grid.txt`, `r05_timeline.txt`, `r05_top_kernels.json`, `r05_roofline_leg_launches.txt` | the bench command under the kernel trace: {first_lines('r05_timeline.txt', 'last run', 1)} | `rocprofv3 --kernel-trace --stats … -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lowerings --no-config4` |",
        "| `r05_experiments.txt` | what was measured on the way and how it came out: the budgets before any change, launch-shape sweeps, key-ordered items, twiddle pairs (COLS tiles, inverse single-crossing passes, LDS twiddle tables), the 9-input mod-down, lazy sums, what was not kept | — |",
        "| `r05_streams16_top_kernels.txt` | the headline program with 16 images per run() in one VM, kernel trace of one run: which kernels the fed regime spends its time in (the counter passes did not survive this run: durations only) | `rocprofv3 --kernel-trace -- python3 tools/legs/headline_only.py 3 --streams 16` |",
        "| `r05_chain_latency.txt`, `r05_profiled_SEAL_MI355X.json` | as in round 4 on this build (the streams table moved into the bench line: `streams`) | `tools/legs/chain_bench.py`, `tools/legs/profile_backend.py` |"]
print("\n".join(r for r in rows if r))
print()
print("## Round 4 (everything `r04_*`; one `gpurun` call of `tools/collect_profiles.sh r04` on the committed build — the JSON files that `bench.py` reads carry "
      "the library's sha256)\n")
print("| file | what (figures read from the file) | command |\n|---|---|---|")
rows = [bench_row("r04"), traffic_row("r04"), step_row("r04"), valu_row("r04"),
        f"| `r04_hybrid_ks_kernels.txt` | one grouped-digit rotation hop at N = 2^17, level 31, kernel by kernel with MEASURED HBM bytes per hop (FETCH_SIZE × 2 + WRITE_SIZE) for the "
        f"three launch sequences, the matrix-core counters, and all levels under HIP events (levels 1 / 7 / 12 / 14 / 31): {hop_levels('r04_hybrid_ks_kernels.txt')}. "
        f"Per hop under the trace: {seq_totals('r04_hybrid_ks_kernels.txt')} | "
        "`tools/collect_profiles.sh` B4: three passes of `tools/legs/hybrid_ks_bench.py 17 39 8 7 10 31 --opt hyb_fuse=f`, `tools/summarize/kernel_bytes.py` |",
        f"| `r04_config4_kernel_stats.csv`, `r04_config4_under_profiler.txt` | BASELINE config 4 under the kernel trace: {stats('r04_config4_kernel_stats.csv')} | "
        "`rocprofv3 --kernel-trace --stats -- python3 tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 8 7` |",
        f"| `r04_boot_kernel_bytes.txt` | ONE real bootstrap at config 4's geometry (38 of them are about nine tenths of config 4), per kernel the measured HBM bytes and GB/s -- the n-ary sums "
        f"and the inner products among them: {first_lines('r04_boot_kernel_bytes.txt', 'bootstrap:', 1)}; {kb_rows('r04_boot_kernel_bytes.txt', 'hyb_mac_kernel<0>|b_sum_group_kernel|b_sum_pair_kernel|ntt_phase_kernel<8, 3, true, false, false>', 4)} "
        "(the process = key generation + encoding + 3 runs).  The counter passes run on one bootstrap because rocprofv3's counter mode "
        "crashes or hangs on the whole config-4 program (`r04_experiments.txt` item 12) | three passes of `tools/legs/boot_demo.py 17 5 1 14 8 7` (`--pmc FETCH_SIZE` / `WRITE_SIZE` with `--opt plan_graph=0`), `tools/summarize/kernel_bytes.py` |",
        f"| `r04_dag_width.txt` | the headline program's dataflow graph: {first_lines('r04_dag_width.txt', 'waves,')}; replay times: {first_lines('r04_dag_width.txt', 'graph replay', 4)} | `python tools/experiments/dag_width.py` |",
        f"| `r04_per_op.json`, `r04_per_op_kernel_stats.csv` | the three expensive opcodes at 13 primes and config 3, kernel by kernel: {stats('r04_per_op_kernel_stats.csv', 3)} | `rocprofv3 --kernel-trace --stats -- python3 tools/legs/per_op_only.py 20` |",
        f"| `r04_kernel_stats.csv`, `r04_by_kernel_and_grid.txt`, `r04_timeline.txt`, `r04_top_kernels.json` | the bench command under the kernel trace: {first_lines('r04_timeline.txt', 'last run', 1)} | `rocprofv3 --kernel-trace --stats … -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lowerings --no-config4` |",
        "| `r04_experiments.txt` | what was measured on the way and how it came out: the fused sequence's first versions, tile-geometry sweeps per ring, loader vs matrix-core conversions, the explicit graph, mixed chains, bounded key sets | — |",
        "| `r04_ntt_full_check.txt`, `r04_roofline_leg_launches.txt`, `r04_chain_latency.txt`, `r04_streams.txt`, `r04_profiled_SEAL_MI355X.json` | as in round 3 on this build | `tools/legs/ntt_full_check.py`, `tools/summarize/summarize_trace.py`, `tools/legs/chain_bench.py`, `bench.py --streams S`, `tools/legs/profile_backend.py` |"]
print("\n".join(r for r in rows if r))
print("\n## Round 3 (everything `r03_*`; `tools/collect_profiles.sh r03` on round 3's committed build)\n")
print("| file | what (figures read from the file) | command |\n|---|---|---|")
rows = [bench_row("r03"), traffic_row("r03"), step_row("r03"), valu_row("r03"),
        f"| `r03_config4_kernel_stats.csv`, `r03_config4_under_profiler.txt` | config 4 on round 3's build: {stats('r03_config4_kernel_stats.csv')} | `rocprofv3 --kernel-trace --stats -- python3 tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 8 7` |",
        "| `r03_step_kernels.json` (note) | its `traffic_over_algorithmic` = 1.52 for `f_ks_frows_mac_kernel<8, 2, 0, true>` is an accounting error of round 3's `tools/summarize/kernel_traffic.py` (the merged form's grid has ℓ + 1 rows; the tool priced it one level too low): fixed in round 4, see `r04_step_kernels.json` | — |",
        "| `r03_hybrid_ks_kernels.txt`, `r03_ntt_full.txt`, `r03_ntt_full_check.txt`, `r03_experiments.txt`, `r03_roofline_leg_launches.txt`, `r03_kernel_stats.csv`, `r03_by_kernel_and_grid.txt`, `r03_timeline.txt`, `r03_top_kernels.json`, `r03_chain_latency.txt`, `r03_streams.txt`, `r03_profiled_SEAL_MI355X.json`, `r03_bench_rccl_world1_broadcast_keys.json` | round 3's records of the grouped-digit hop (matrix-core vs vector conversions), the single-crossing NTT (ablations, counters, variants that lost), its experiments log, traces, chain latency, streams, the per-op table in the reference compiler's schema, the RCCL path at world size 1 | see each file's header |"]
print("\n".join(r for r in rows if r))
hist = P / "README.history.md"
if hist.exists():
    print("\n" + hist.read_text().rstrip())
