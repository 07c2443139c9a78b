#!/usr/bin/env python3
"""Runs only the batched forward/inverse NTT (for counter collection): python3 tools/legs/ntt_only.py [logN] [limbs] [iters]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from dacapo_amd import lowlevel as ll

logN = int(sys.argv[1]) if len(sys.argv) > 1 else 15
limbs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = ll.Context(logN, 14)
buf = ll.DeviceBuffer((limbs, 1 << logN))
ll.lib().dc_memset(buf.ptr, 1, buf.nbytes)
for _ in range(iters):
    ctx.ntt(buf, limbs, prime_base=0, prime_period=14)
    ctx.ntt(buf, limbs, inverse=True, prime_base=0, prime_period=14)
ctx.sync()
