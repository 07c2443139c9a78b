#!/usr/bin/env python3
"""Headline program only (ResNet-20 fixture): ms per run() and the decrypted error; for kernel-tuning sweeps.
    [DACAPO_AMD_LIB=...] python tools/experiments/quick_headline.py [steps] [fixture = resnet20 | resnet20.b6 | resnet20.b13]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import runner  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
fx = ha.read_fixture(ROOT / "tests" / "golden" / "resnet20")
if len(sys.argv) > 2 and sys.argv[2] != "resnet20":  # another lowering of the same trace: same constants, other bytecode
    import gzip
    fx["hevm"] = gzip.open(ROOT / "tests" / "golden" / (sys.argv[2] + ".hevm.gz")).read()
vm = runner.HEVM(fresh=True, logN=15, num_primes=14)
vm.load_mem(fx["cst"], fx["hevm"])
vm.setInput(0, fx["packed"])
vm.run()
best, tot = 1e9, 0.0
for _ in range(steps):
    t0 = time.perf_counter()
    vm.run()
    dt = time.perf_counter() - t0
    best, tot = min(best, dt), tot + dt
out = vm.getOutput()[0]
print(json.dumps({"ms_avg": round(tot / steps * 1e3, 2), "ms_min": round(best * 1e3, 2),
                  "rms_vs_torch": float(np.sqrt(np.mean((out[:10] * 32 - fx["torch_result"]) ** 2)))}))
