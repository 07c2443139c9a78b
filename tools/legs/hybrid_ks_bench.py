#!/usr/bin/env python3
"""One rotation hop with grouped-digit hybrid key switching at N = 2^17 (hybrid_ks.hip), level by level: HIP-event time per hop and the
algorithmic bytes of each kernel of the sequence, so that a `rocprofv3 --kernel-trace --stats` of this tool (summarised by
tools/experiments/hybrid_ks_summary.py) places every kernel on the byte roofline.   python3 tools/legs/hybrid_ks_bench.py [logN=17] [K=39] [ks=8] [alpha=7] [iters=5] [only_level=0] [--opt name=value ...]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from dacapo_amd import lowlevel as ll  # noqa: E402
from dacapo_amd import runner  # noqa: E402

sys.argv = runner.apply_cli_options(sys.argv)  # --opt name=value (csrc/options.hpp), e.g. --opt hyb_fuse=0

logN = int(sys.argv[1]) if len(sys.argv) > 1 else 17
K = int(sys.argv[2]) if len(sys.argv) > 2 else 39
ks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
alpha = int(sys.argv[4]) if len(sys.argv) > 4 else 7
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
only = int(sys.argv[6]) if len(sys.argv) > 6 else 0
ctx = ll.Context(logN, K, special=ks, alpha=alpha)
L, N = ll.lib(), 1 << logN
Lmax, D = ctx.max_level, ctx.key_digits
key = ll.DeviceBuffer((D, 2, K, N))
L.dc_memset(key.ptr, 3, key.nbytes)
e0, e1 = L.dc_event_create(), L.dc_event_create()
p_limb = 8 * N
rows = []
for ell in ([only] if only else sorted({1, alpha, 12, 14, Lmax} & set(range(1, Lmax + 1)) | {Lmax})):
    a, d = ll.DeviceBuffer((2, ell, N)), ll.DeviceBuffer((2, ell, N))
    L.dc_memset(a.ptr, 1, a.nbytes)
    st = ell * N
    for _ in range(2):
        L.dc_ct_rotate_hop(ctx.h, d.ptr, st, a.ptr, st, 3, key.ptr, ell, None)
    L.dc_event_record(e0, None)
    for _ in range(iters):
        L.dc_ct_rotate_hop(ctx.h, d.ptr, st, a.ptr, st, 3, key.ptr, ell, None)
    L.dc_event_record(e1, None)
    us = L.dc_event_elapsed_ms(e0, e1) / iters * 1e3
    G, M = -(-ell // alpha), ell + ks
    E = G * M - ell
    alg = {"prepare": 4 * ell, "modup": ell + E, "mac": E + ell + 2 * G * M + 2 * M, "moddown": 2 * ks + 2 * ell, "final": 2 * ell + 2 * ell + ell + 2 * ell}
    ntts = G * M + 2 * ks + 2 * ell
    total = (2 * ell + 2 * G * M + 2 * ell) + sum(alg.values())  # + transforms: 2 limbs each (in place), the key counted once (mac)
    rows.append({"level": ell, "digits": G, "hop_us": round(us, 1), "ntt_equivalents": ntts, "ntt_per_s": round(ntts / (us * 1e-6)),
                 "limbs_by_kernel": alg, "algorithmic_bytes": total * p_limb, "achieved_gbs": round(total * p_limb / (us * 1e-6) / 1e9, 1),
                 "frac_of_hbm_peak": round(total * p_limb / (us * 1e-6) / 8e12, 4),
                 "seal_scheme_ntt_equivalents": (ell + 1) * (ell + 2)})
    del a, d
print(json.dumps({"N": N, "primes": K, "special": ks, "alpha": alpha, "key_bytes": key.nbytes, "levels": rows}))
