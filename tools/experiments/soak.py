#!/usr/bin/env python3
"""Soak: the headline program run() `n` times in one VM -- time per run, decrypted error and free HBM must not drift.
    python tools/experiments/soak.py [n=300]"""
import ctypes
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import lowlevel as ll  # noqa: E402
from dacapo_amd import runner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
L = ll.lib()
L.dc_mem_info.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
L.dc_mem_info.restype = None


def free_bytes():
    f, t = ctypes.c_uint64(), ctypes.c_uint64()
    L.dc_mem_info(ctypes.byref(f), ctypes.byref(t))
    return f.value


fx = ha.read_fixture(ROOT / "tests" / "golden" / "resnet20")
vm = runner.HEVM(fresh=True, logN=15, num_primes=14)
vm.load_mem(fx["cst"], fx["hevm"])
vm.setInput(0, fx["packed"])
vm.run()
f0 = free_bytes()
ts, rms = [], []
for i in range(n):
    t0 = time.perf_counter()
    vm.run()
    ts.append(time.perf_counter() - t0)
    if i % 50 == 49:
        out = vm.getOutput()[0]
        rms.append(float(np.sqrt(np.mean((out[:10] * 32 - fx["torch_result"]) ** 2))))
f1 = free_bytes()
ts = np.array(ts) * 1e3
print(json.dumps({"runs": n, "ms_first_50": round(float(ts[:50].mean()), 3), "ms_last_50": round(float(ts[-50:].mean()), 3),
                  "ms_min": round(float(ts.min()), 3), "ms_max": round(float(ts.max()), 3), "rms_vs_torch_every_50": [round(r, 6) for r in rms],
                  "free_hbm_change_bytes": int(f1) - int(f0)}))
