import os
import sys
import tempfile
os.environ.setdefault("DACAPO_AMD_HOOKS", "1")  # seeded keys: the hooks build (csrc/test_hooks.hip)
from pathlib import Path
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from dacapo_amd import ckks_boot as cb, hevm_asm as ha, lowlevel as ll, runner
from gpu_helpers import _get_ct, _import_keys, _mirror_vm
from oracle.oracle import Oracle
logN, ks = 12, 1
K0 = 3 + cb.boot_levels() + ks
primes = cb.mixed_prime_chain(logN, [60] + [51] * (K0 - 1 - ks) + [60] * ks)
b = ha.Builder(slots=1 << (logN - 1), init_level=1, shadow=False)
x = b.input(None, level=1, scale_bits=40)
em = cb.BootstrapEmitter(b, logN, K0, 3, ks=ks, primes=primes)
y, _ = em.bootstrap(x, 2.0**40); b.output(y)
cst, hv, info = b.assemble()
h = ha.unpack_hevm(hv); ops = h["ops"]
offs = cb.rotation_offsets(hv)
hevm = runner.HEVM(seed=21, logN=logN, num_primes=K0, vm_options={"plan": 0, "secret_hw": 32}, ks_special=ks, primes=primes)
hevm.addRotationKeys(offs)
o = Oracle(logN, K0, primes=primes)
_import_keys(o, hevm, ll)
for step in offs:
    elt = o.elt_from_step(step)
    o.galois[elt] = ll.read_device(hevm.lw.hevm_galois_key(hevm.vm, elt), (K0 - 1, 2, K0, o.N))
names = {v: k for k, v in vars(ha).items() if k.startswith('OP_') and isinstance(v, int)}
msg = np.random.default_rng(8).uniform(-1, 1, o.slots)
def run(n):
    sub = ops[:n]
    last = [int(d) for opc, d, l, r in sub.tolist() if opc not in (ha.OP_ENCODE, ha.OP_ENCODE_COMPLEX)][-1]
    hv2 = ha.pack_hevm(h["arg_scale"], h["arg_level"], [40], [1], [last], h["num_ctxt"], h["num_ptxt"], h["init_level"], sub)
    hevm.load_mem(cst, hv2)
    tmp = Path(tempfile.mkdtemp())
    ovm = _mirror_vm(hevm, ll, o, cst, hv2, tmp)
    hevm.setInput(0, msg)
    ovm.ciphers[0] = _get_ct(hevm, ll, 0)
    hevm.run(); ovm.run()
    got, want = _get_ct(hevm, ll, last), ovm.ciphers[last]
    return got.ell == want.ell and got.scale == want.scale and bool((got.data == want.data).all()), got, want
ct_idx = [i for i, (opc, d, l, r) in enumerate(ops.tolist()) if opc not in (ha.OP_ENCODE, ha.OP_ENCODE_COMPLEX)]
lo, hi = 0, len(ct_idx) - 1      # find the first ct instruction after which results differ
assert not run(ct_idx[hi] + 1)[0], "full program agrees?!"
while lo < hi:
    mid = (lo + hi) // 2
    ok = run(ct_idx[mid] + 1)[0]
    if ok: lo = mid + 1
    else: hi = mid
n = ct_idx[lo]
opc, d, l, r = ops[n].tolist()
ok, got, want = run(n + 1)
print("first differing instruction index", n, names.get(opc, opc), "dst", d, "lhs", l, "rhs", r, "level", got.ell, want.ell, "scale", got.scale, want.scale)
diff = np.argwhere(got.data != want.data)
print("mismatching entries", len(diff), "of", got.data.size, "first", diff[:5].tolist())
for (p, i, k) in diff[:3].tolist():
    print("  poly", p, "limb", i, "coef", k, "gpu", int(got.data[p, i, k]), "oracle", int(want.data[p, i, k]), "q", primes[i])
# who is right?  recompute the first mismatches of a ct x pt product with Python integers
if opc == ha.OP_MULCP:
    import ctypes
    src = _get_ct(hevm, ll, l)
    lvl_, sc_ = ctypes.c_int32(), ctypes.c_double()
    pp = hevm.lw.hevm_plain(hevm.vm, r, ctypes.byref(lvl_), ctypes.byref(sc_))
    pt = ll.read_device(pp, (lvl_.value, o.N))
    for (p, i, k) in diff[:6].tolist():
        a, bb, q = int(src.data[p, i, k]), int(pt[i, k]), primes[i]
        print("  a", a, "b", bb, "q", q, "a<q", a < q, "b<q", bb < q, "a*b%q", (a * bb) % q, "gpu ok", (a * bb) % q == int(got.data[p, i, k]), "oracle ok", (a * bb) % q == int(want.data[p, i, k]))
# does the GPU's source register change between its definition and this use?
ref = None
for m in range(6, n + 2):
    if ops[m - 1][0] in (ha.OP_ENCODE, ha.OP_ENCODE_COMPLEX):
        continue
    okm, g_, w_ = run(m)
    cur = _get_ct(hevm, ll, l)
    if ref is None:
        ref = cur
    same_gpu = bool(cur.ell == ref.ell and (cur.data == ref.data).all())
    print("prefix", m, names.get(int(ops[m - 1][0])), "dst", int(ops[m - 1][1]), "prefix result ok", okm, "| GPU reg", l, "unchanged", same_gpu)
