#!/usr/bin/env python3
"""Places the kernels of one grouped-digit key switch on the byte roofline:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/legs/hybrid_ks_bench.py 17 39 8 7 5 31 > hop.json
    python tools/experiments/hybrid_ks_summary.py out/*/*kernel_stats.csv hop.json
per kernel of the sequence: average duration, algorithmic bytes (limbs x 8 N, tools/legs/hybrid_ks_bench.py), GB/s, fraction of 8 TB/s."""
import csv
import json
import re
import sys

stats = {re.sub(r"\(.*", "", r["Name"]).replace("void dacapo::", "").replace("dacapo::", ""): r for r in csv.DictReader(open(sys.argv[1]))}
hop = json.load(open(sys.argv[2]))
row = hop["levels"][-1]
p_limb = 8 * hop["N"]
G, ell, ks = row["digits"], row["level"], hop["special"]
M = ell + ks
limbs = dict(row["limbs_by_kernel"])
names = {"prepare": "hyb_prepare_rot_kernel", "modup": "hyb_modup_kernel", "mac": "hyb_mac_kernel<0>", "moddown": "hyb_moddown_kernel", "final": "hyb_final_kernel<0>"}
if any(k.startswith("hyb_conv_mfma_kernel") for k in stats):  # the matrix-core form of the two conversions
    names["modup"], names["moddown"] = "hyb_conv_mfma_kernel<false>", "hyb_conv_mfma_kernel<true>"
print(f"N = {hop['N']}, level {ell}, {G} digits of {hop['alpha']} primes, {ks} special primes: hop {row['hop_us']} us under HIP events, "
      f"{row['ntt_equivalents']} NTT-equivalents (SEAL's scheme at this level: {row['seal_scheme_ntt_equivalents']})")
print(f"{'kernel':42s} {'calls':>6s} {'avg us':>9s} {'limbs':>6s} {'MB':>8s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
tot = 0.0
for k, n in names.items():
    r = next((v for kk, v in stats.items() if kk.startswith(n)), None)
    if not r:
        continue
    us = float(r["AverageNs"]) / 1e3
    b = limbs[k] * p_limb
    tot += us
    print(f"{n:42s} {r['Calls']:>6s} {us:9.1f} {limbs[k]:6d} {b/1e6:8.1f} {b/us/1e3:8.1f} {b/us/1e3/8000:9.3f}")
ntt_us = 0.0
for kk, r in stats.items():
    if kk.startswith("ntt_"):
        hops = next((int(v["Calls"]) for kk2, v in stats.items() if kk2.startswith(names["prepare"])), 1)
        us = float(r["TotalDurationNs"]) / 1e3 / hops
        ntt_us += us
        print(f"{kk[:42]:42s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f}   (transforms: {us:.1f} us per hop)")
print(f"element-wise kernels {tot:.1f} us + transforms {ntt_us:.1f} us per hop")
