import os
import sys
os.environ.setdefault("DACAPO_AMD_HOOKS", "1")  # seeded keys: the hooks build (csrc/test_hooks.hip)
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from dacapo_amd import ckks_boot as cb, lowlevel as ll, runner
from gpu_helpers import _import_keys
from oracle.oracle import Oracle, Ciphertext, splitmix_fill
logN, ks = 12, 1
K = 3 + cb.boot_levels() + ks
primes = cb.mixed_prime_chain(logN, [60] + [51] * (K - 1 - ks) + [60] * ks)
hevm = runner.HEVM(seed=21, logN=logN, num_primes=K, vm_options={"plan": 0, "secret_hw": 32}, ks_special=ks, primes=primes)
o = Oracle(logN, K, primes=primes)
_import_keys(o, hevm, ll)
pr = np.array(primes, dtype=np.uint64)
elt = o.elt_from_step(1)
key = o.galois[elt]
print("key canonical:", bool((key < pr[None, None, :, None]).all()), "sk canonical", bool((o.sk < pr[:, None]).all()), "pk canonical", bool((o.pk < pr[None, :, None]).all()))
bad = np.argwhere(key >= pr[None, None, :, None])
print("non-canonical key entries:", len(bad), bad[:5].tolist())
N = 1 << logN
ctxh = hevm.ctx_handle
L = ll.lib_gw()
for ell in (5, 16, 17, 20):
    q = pr[:ell, None]
    a = np.stack([np.stack([splitmix_fill(1 + 7 * p + i + 100 * ell, N) for i in range(ell)]) % q for p in range(2)])
    da, dd = ll.DeviceBuffer.from_host(a), ll.DeviceBuffer((2, ell, N))
    dk = ll.DeviceBuffer.from_host(key)
    L.dc_ct_rotate_hop(ctxh, dd.ptr, ell * N, da.ptr, ell * N, elt, dk.ptr, ell, None)
    got = dd.to_host()
    want = o.apply_galois(Ciphertext(a, 2.0**40), elt).data
    print("level", ell, "C-ABI hop with the VM's key == oracle:", bool((got == want).all()), "mismatches", int((got != want).sum()))
