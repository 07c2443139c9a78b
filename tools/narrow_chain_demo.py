#!/usr/bin/env python3
"""A whole VM program on a chain of 51-bit primes (the width of the reference's HEaaN rescale primes, profiled_HEAAN_GPU.json
rescalingFactor 51) -- on the generic-width build of the library -- against the oracle VM on the same primes, keys, plaintexts and input:
final ciphertext limbs and the decrypted values.  Run with the environment of the generic-width build:
    DACAPO_AMD_LIB=dacapo_amd/lib/libSEAL_HEVM_gw.so DACAPO_HEVM_PRIME_BITS=51 python tools/narrow_chain_demo.py [logN=12] [K=7]
(test infrastructure: the oracle is the checker; tests/test_gpu_prime_widths.py runs this as a child process)"""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
bits = int(os.environ.get("DACAPO_HEVM_PRIME_BITS", "60"))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import lowlevel as ll  # noqa: E402
from dacapo_amd import runner  # noqa: E402
from gpu_helpers import _get_ct, _import_keys, _mirror_vm  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

logN = int(sys.argv[1]) if len(sys.argv) > 1 else 12
K = int(sys.argv[2]) if len(sys.argv) > 2 else 7
slots = 1 << (logN - 1)
rng = np.random.default_rng(6)
# ciphertexts at 2^40, plaintexts at one prime's worth of scale (2^bits), one rescale per product: the lazy policy with `bits`-bit primes
b = ha.Builder(slots=slots, init_level=K - 1, policy="lazy", boot_level=K - 1, rescale_bits=bits, shadow=True)
x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
t = b.add(b.mul(x, y), b.rotate(x, 3))
t = b.add(b.mul_plain(t, rng.uniform(-1, 1, slots)), b.rotate(y, -5))
u = b.mul(t, t)
u = b.add(u, b.rotate(b.mul_plain(x, [0.25]), 33))
b.output(b.finish(u))
cst, hv, info = b.assemble()
hevm = runner.HEVM(seed=77, logN=logN, num_primes=K)
o = Oracle(logN, K, bit_size=bits)
assert [int(p).bit_length() for p in o.primes] == [bits] * K
_import_keys(o, hevm, ll)
hevm.load_mem(cst, hv)
tmp = Path(tempfile.mkdtemp())
ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp)
for i, a in enumerate(b.args):
    hevm.setInput(i, a.plain)
    ovm.ciphers[i] = _get_ct(hevm, ll, i)
hevm.run()
ovm.run()
r = ovm.prog.res_dst[0]
got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
out = hevm.getOutput()[0]
print(json.dumps({"prime_bits": bits, "primes": [hex(p) for p in o.primes], "levels_left": int(got.ell), "limbs_identical": bool(got.ell == want.ell and (got.data == want.data).all()),
                  "scale_identical": bool(got.scale == want.scale), "max_error_vs_cleartext": float(np.abs(out - b.expected()[0]).max()), "op_mix": info["op_mix"]}))
