#!/usr/bin/env python3
"""The single-crossing NTT (ntt_full.hip) against the two-launch tiles on the GPU: bit-exactness (forward, inverse, round trip) and the
time of both over `limbs` limbs of N = 2^15.   python tools/legs/ntt_full_check.py [limbs=4096] [iters=10] [--opt name=value ...]"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from dacapo_amd import lowlevel as ll  # noqa: E402
from dacapo_amd import runner  # noqa: E402

sys.argv = runner.apply_cli_options(sys.argv)  # --opt name=value (e.g. ntt_full_inv_pairs=0)

limbs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = ll.Context(15, 14)
L, N, K = ll.lib(), ctx.N, ctx.K
n_chk = min(limbs, 56)
q = np.array([ctx.primes[i % K] for i in range(n_chk)], dtype=np.uint64)[:, None]
host = (np.random.default_rng(1).integers(0, 1 << 62, size=(n_chk, N), dtype=np.uint64)) % q
res = {}
for inv in (False, True):
    out = []
    for variant in (0, 1):
        buf = ll.DeviceBuffer.from_host(host)
        ctx.ntt(buf, n_chk, inverse=inv, prime_base=0, prime_period=K, variant=variant)
        out.append(buf.to_host())
    res["inverse" if inv else "forward"] = bool((out[0] == out[1]).all())
    if not res["inverse" if inv else "forward"]:
        bad = np.argwhere(out[0] != out[1])
        print("first mismatches", bad[:8].tolist(), file=sys.stderr)
buf = ll.DeviceBuffer.from_host(host)
ctx.ntt(buf, n_chk, prime_base=0, prime_period=K, variant=1)
ctx.ntt(buf, n_chk, inverse=True, prime_base=0, prime_period=K, variant=1)
res["round_trip"] = bool((buf.to_host() == host).all())
big = ll.DeviceBuffer((limbs, N))
fill = (np.arange(limbs * N, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(5)
L.dc_memcpy_h2d(big.ptr, fill.ctypes.data, fill.nbytes)
e0, e1 = L.dc_event_create(), L.dc_event_create()
for inv in (False, True):
    for variant in (0, 1):
        for _ in range(2):
            ctx.ntt(big, limbs, inverse=inv, prime_base=0, prime_period=K, variant=variant)
        L.dc_event_record(e0, None)
        for _ in range(iters):
            ctx.ntt(big, limbs, inverse=inv, prime_base=0, prime_period=K, variant=variant)
        L.dc_event_record(e1, None)
        us = L.dc_event_elapsed_ms(e0, e1) / iters * 1e3
        res[("inv" if inv else "fwd") + ("_full_us" if variant else "_two_phase_us")] = round(us, 1)
        res[("inv" if inv else "fwd") + ("_full_frac" if variant else "_two_phase_frac")] = round(2.0 * limbs * N * 8 / (us * 1e-6) / 8e12, 4)
res["limbs"] = limbs
print(json.dumps(res))
