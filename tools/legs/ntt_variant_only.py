#!/usr/bin/env python3
"""Runs one NTT implementation alone, for rocprofv3 (kernel trace / --pmc): python3 tools/legs/ntt_variant_only.py variant [limbs=4096] [iters=3]
variant 0 = two-launch tiles, 1 = single-crossing kernel (dc_ntt_variant); forward and inverse alternate on N = 2^15."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from dacapo_amd import lowlevel as ll

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
limbs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = ll.Context(15, 14)
buf = ll.DeviceBuffer((limbs, 1 << 15))
ll.lib().dc_memset(buf.ptr, 1, buf.nbytes)
for _ in range(iters):
    ctx.ntt(buf, limbs, prime_base=0, prime_period=14, variant=variant)
    ctx.ntt(buf, limbs, inverse=True, prime_base=0, prime_period=14, variant=variant)
ctx.sync()
