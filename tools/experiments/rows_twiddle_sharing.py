#!/usr/bin/env python3
"""Probe: forward two-launch transform of 160 limbs at N = 2^17 when the limbs of one prime are 32 limbs apart (the raised limbs' layout:
digit by digit) against adjacent (prime by prime).  Under rocprofv3 --kernel-trace --stats the ROWS phase's time says whether sharing a
prime's twiddles in L2 pays.   python3 tools/experiments/rows_twiddle_sharing.py [apart|adjacent] [iters]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from dacapo_amd import lowlevel as ll

mode = sys.argv[1] if len(sys.argv) > 1 else "apart"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
logN, limbs = 17, 160
ctx = ll.Context(logN, 39)
buf = ll.DeviceBuffer((limbs, 1 << logN))
ll.lib().dc_memset(buf.ptr, 1, buf.nbytes)
idx = np.array([(r % 32) if mode == "apart" else (r // 5) for r in range(limbs)], dtype=np.int32)
d_idx = ll.DeviceBuffer.from_host(idx)
for _ in range(iters):
    ctx.ntt(buf, limbs, prime_idx=d_idx, prime_period=limbs)
ctx.sync()
