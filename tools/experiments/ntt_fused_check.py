#!/usr/bin/env python3
"""Large-batch NTT (the throughput geometry) against the CPU oracle, bit for bit, + round trips; written for the single-launch
experiment (r02_ntt_fused_xcd_cluster.patch, DACAPO_NTT_FUSED=1 there), valid for any build:
    python tools/experiments/ntt_fused_check.py [logN=15] [limbs=520] [K=14]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from dacapo_amd import lowlevel as ll  # noqa: E402
from oracle.oracle import Oracle, splitmix_fill  # noqa: E402

logN = int(sys.argv[1]) if len(sys.argv) > 1 else 15
limbs = int(sys.argv[2]) if len(sys.argv) > 2 else 520
K = int(sys.argv[3]) if len(sys.argv) > 3 else 14
ctx, o = ll.Context(logN, K), Oracle(logN, K)
N = 1 << logN
pidx = [b % K for b in range(limbs)]
a = np.stack([splitmix_fill(0x4845564D + b, N) % np.uint64(o.primes[p]) for b, p in enumerate(pidx)])
d = ll.DeviceBuffer.from_host(a)
for rep in range(3):  # several launches: the cluster counters are monotone across launches
    ctx.ntt(d, limbs, prime_base=0, prime_period=K)
    got = d.to_host()
    if rep == 0:
        t0 = time.time()
        check = sorted(set(list(range(0, limbs, max(1, limbs // 40))) + [limbs - 1, limbs - 2, 7, 8]))
        want = o.ntt_fwd(a[check], [pidx[b] for b in check])
        print(f"oracle forward NTT of {len(check)} limbs: {time.time()-t0:.1f} s")
        first = got.copy()
    assert (got[check] == want).all(), "forward mismatch"
    assert (got == first).all(), "launches disagree"
    ctx.ntt(d, limbs, inverse=True, prime_base=0, prime_period=K)
    back = d.to_host()
    assert (back == a).all(), "round trip mismatch"
print(f"N=2^{logN}, {limbs} limbs: forward == oracle on {len(check)} limbs, 3 x (forward, inverse) round trips exact")
