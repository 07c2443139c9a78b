#!/usr/bin/env python3
"""The reference's ResNet-20 with REAL CKKS bootstrapping at every bootstrap site (the headline program tests/golden/resnet20.*, every
opcode 10 rewritten by dacapo_amd/ckks_boot.lower_bootstraps): BASELINE config 4 in spirit -- the reference
runs it on HEaaN (HEAAN_HEVM.cpp:386-399) at N = 2^17; here on SEAL-style 60-bit primes, N = 2^15, 20 primes, sparse secret.
    python tools/resnet_real_boot.py [direct_keys=1]"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("DACAPO_HEVM_SECRET_HW", "64")
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import ckks_boot as cb  # noqa: E402
from dacapo_amd import runner  # noqa: E402

direct = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fx = ha.read_fixture(ROOT / "tests" / "golden" / "resnet20")
t0 = time.time()
fx["hevm"], fx["cst"] = cb.lower_bootstraps(fx["hevm"], fx["cst"], 15, 20, msg_bits=4)
print(f"opcode 10 -> real bootstrapping: {time.time()-t0:.1f} s", flush=True)
h = ha.unpack_hevm(fx["hevm"])
ops = h["ops"]
print(f"{len(ops)} instructions, {h['num_ptxt']} plaintext registers, {int((ops[:, 0] == ha.OP_MODRAISE).sum())} real bootstraps", flush=True)
t0 = time.time()
hevm = runner.HEVM(seed=0x4845564D, logN=15, num_primes=20)
if direct:
    offs = sorted({(int(q) - 65536 if q >= 32768 else int(q)) for o, _, _, q in ops.tolist() if o == ha.OP_ROTATE} - {0})
    hevm.addRotationKeys(offs)
    print(f"{len(offs)} direct rotation keys", flush=True)
print(f"context + keys {time.time()-t0:.1f} s", flush=True)
t0 = time.time()
hevm.load_mem(fx["cst"], fx["hevm"])
print(f"load + preprocess (encode, plan, graph) {time.time()-t0:.1f} s", flush=True)
hevm.setInput(0, fx["packed"])
t0 = time.perf_counter()
hevm.run()
dt = time.perf_counter() - t0
out = hevm.getOutput()[0]
st = hevm.stats()
res = {"run_s": round(dt, 3), "key_switches": st["keyswitches"], "ntt_equivalents": st["ntts"], "ntt_per_s": round(st["ntts"] / dt),
       "real_bootstraps": int((ops[:, 0] == ha.OP_MODRAISE).sum()),
       "rms_vs_torch": float(np.sqrt(np.mean((out[:10] * 32 - fx["torch_result"]) ** 2))),
       "rms_vs_plaintext_evaluation": float(np.sqrt(np.mean((out - fx["expected"]) ** 2))),
       "logits": [round(float(v), 4) for v in out[:10] * 32], "torch": [round(float(v), 4) for v in fx["torch_result"]]}
print(json.dumps(res))
