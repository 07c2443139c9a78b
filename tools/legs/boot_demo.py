#!/usr/bin/env python3
"""Real CKKS bootstrapping on the MI355X: one ciphertext at 1 prime -> `target` primes, through the extension opcodes and
dacapo_amd/ckks_boot.py.   python tools/legs/boot_demo.py [logN=15] [r=5] [direct_keys=1] [target=3] [ks_special=1] [ks_alpha=ks_special] [--opt name=value ...]
ks_special > 1: grouped-digit hybrid key switching (hybrid_ks.hip), the chain gets that many special primes.
BASELINE config 4's geometry: python tools/legs/boot_demo.py 17 5 1 14 8 7"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from dacapo_amd import ckks_boot as cb  # noqa: E402
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import runner  # noqa: E402

sys.argv = runner.apply_cli_options(sys.argv)
logN = int(sys.argv[1]) if len(sys.argv) > 1 else 15
r = int(sys.argv[2]) if len(sys.argv) > 2 else 5
direct = int(sys.argv[3]) if len(sys.argv) > 3 else 1
target = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ks = int(sys.argv[5]) if len(sys.argv) > 5 else 1
alpha = int(sys.argv[6]) if len(sys.argv) > 6 else ks
K, cst, hv, offs_all, em = cb.single_bootstrap_program(logN, target=target, r=r, ks=ks)
slots = 1 << (logN - 1)
info = {'num_ops': len(ha.unpack_hevm(hv)['ops']), 'num_ptxt': ha.unpack_hevm(hv)['num_ptxt']}
print(f"N=2^{logN}, {K} primes ({ks} special), target {target}, r={r}: {info['num_ops']} instructions, {info['num_ptxt']} plaintexts, {len(cst)/1e6:.0f} MB of constants")
msg = np.random.default_rng(3).uniform(-1, 1, slots)
sim = cb.simulate(hv, cst, [msg], logN, em.primes)[0]
print("cleartext simulation: max error", np.abs(sim - msg).max())
t0 = time.time()
hevm = runner.HEVM(fresh=True, logN=logN, num_primes=K, ks_special=ks, ks_alpha=alpha, vm_options={"secret_hw": 64})
print(f"context + keys: {time.time()-t0:.1f} s")
if direct:
    offs = offs_all
    t0 = time.time()
    hevm.addRotationKeys(offs)
    print(f"{len(offs)} direct rotation keys: {time.time()-t0:.1f} s")
t0 = time.time()
hevm.load_mem(cst, hv)
print(f"load + preprocess: {time.time()-t0:.1f} s")
hevm.setInput(0, msg)
hevm.run()
for _ in range(2):
    t0 = time.perf_counter()
    hevm.run()
    dt = time.perf_counter() - t0
out = hevm.getOutput()[0]
err = np.abs(out - msg)
st = hevm.stats()
c = hevm.getCtxt(hevm.getResIdx(0))
print(f"bootstrap: {dt*1e3:.2f} ms, {st['keyswitches']} key switches, {st['ntts']} NTT-equivalents, result at {c.level} primes, scale 2^{np.log2(c.scale):.3f}")
print(f"decrypted vs message: max error {err.max():.3e}, rms {np.sqrt(np.mean(err**2)):.3e}  ({-np.log2(err.max()):.1f} bits)")
