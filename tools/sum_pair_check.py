#!/usr/bin/env python3
"""The n-ary sum of ciphertexts and ciphertext x plaintext products (plan: P_SUM, batch_ops.hip b_sum) against the oracle VM, limb for limb, with
either form of the kernel: DACAPO_SUM_PAIR_MIN_WGS=0 forces the paired form (both polynomials in one thread), a huge value the split form.
    DACAPO_SUM_PAIR_MIN_WGS=0 python tools/sum_pair_check.py [logN=13] [K=5]
(test infrastructure: the oracle is the checker; tests/test_gpu_hevm.py runs this as a child process, because the threshold is read once)"""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import lowlevel as ll  # noqa: E402
from dacapo_amd import runner  # noqa: E402
from gpu_helpers import _get_ct, _import_keys, _mirror_vm  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

logN = int(sys.argv[1]) if len(sys.argv) > 1 else 13
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
slots = 1 << (logN - 1)
rng = np.random.default_rng(9)
b = ha.Builder(slots=slots, init_level=K - 1, policy="lazy", boot_level=K - 1, shadow=True)
x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
# a convolution-shaped sum: 19 products with plaintexts (more than one 16-term reduction window) ...
acc = None
for k in range(19):
    t = b.mul_plain(b.rotate(x if k % 3 else y, 1 << (k % 7)), rng.uniform(-1, 1, slots))
    acc = t if acc is None else b.add(acc, t)
# ... plus 9 bare ciphertext terms at the same scale (more than one 8-term lazy window), and one more product level on top
z = b.mul_plain(y, rng.uniform(-1, 1, slots))
for k in range(9):
    acc = b.add(acc, b.rotate(z, 3 + 2 * k))
# (two readers, one of them a rotation: the sum is materialised by the sum kernel instead of being folded into a rescale's loaders)
u = b.add(acc, b.rotate(acc, 5))
b.output(b.finish(b.mul(u, u)))
cst, hv, info = b.assemble()
hevm = runner.HEVM(seed=31, logN=logN, num_primes=K)
o = Oracle(logN, K)
_import_keys(o, hevm, ll)
hevm.load_mem(cst, hv)
ovm = _mirror_vm(hevm, ll, o, cst, hv, Path(tempfile.mkdtemp()))
for i, a in enumerate(b.args):
    hevm.setInput(i, a.plain)
    ovm.ciphers[i] = _get_ct(hevm, ll, i)
hevm.run()
ovm.run()
r = ovm.prog.res_dst[0]
got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
out = hevm.getOutput()[0]
print(json.dumps({"pair_min_wgs": os.environ.get("DACAPO_SUM_PAIR_MIN_WGS"), "limbs_identical": bool(got.ell == want.ell and (got.data == want.data).all()),
                  "scale_identical": bool(got.scale == want.scale), "max_error_vs_cleartext": float(np.abs(out - b.expected()[0]).max()),
                  "op_mix": info["op_mix"], "stats": hevm.stats()}))
