#!/usr/bin/env python3
"""Where a bootstrap's error comes from, measured on the CPU oracle (test infrastructure: this tool is not part of the product path).
One real bootstrap (dacapo_amd/ckks_boot.py) of a random message through oracle.OracleVM; the error is then taken per COEFFICIENT of
the decrypted polynomials (a bootstrap works on coefficients; a slot error is sqrt(N) times the coefficient error) and tabulated by
the ModRaise overflow I of that coefficient: noise entering before EvalMod's double angles is amplified by 2^r / sin(theta(I)) and shows
up as a dependence on I; noise of the linear transforms does not.  Round 3 used it to find the three dominant terms (key-switch
noise of the baby-step rotations, matrix-plaintext rounding, the conjugation's key switch; ckks_boot.BootstrapEmitter.bootstrap).
    python tools/experiments/boot_precision.py logN secret_weight [msg_bits=0] [amplitude=1]      (N = 2^13: 20 s, N = 2^15: 2 min)"""
import sys, time, os, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from dacapo_amd import ckks_boot as cb, hevm_asm as ha
from oracle.oracle import Oracle, OracleVM
logN = int(sys.argv[1]); h = int(sys.argv[2]); mb = int(sys.argv[3]) if len(sys.argv)>3 else 0
amp = float(sys.argv[4]) if len(sys.argv)>4 else 1.0
K, cst, hv, offs, em = cb.single_bootstrap_program(logN, msg_bits=mb)
print("K", K)
o = Oracle(logN, K)
t=time.time()
o.keygen_sparse(h, seed=3, galois_elts=sorted(set(o.default_galois_elts()) | {o.elt_from_step(s) for s in offs}))
print("keygen", time.time()-t)
tmp = Path(tempfile.mkdtemp())
(tmp / 'p.cst').write_bytes(cst); (tmp / 'p.hevm').write_bytes(hv)
vm = OracleVM(o); vm.load(tmp / 'p.cst', tmp / 'p.hevm'); vm.preprocess()
msg = np.random.default_rng(1).uniform(-amp, amp, o.slots)
vm.encrypt(0, msg)
t=time.time(); vm.run(); print("run", time.time()-t)
e = np.abs(vm.decrypt_result(0)-msg)
print(f"logN={logN} h={h} mb={mb} amp={amp}: max {e.max():.3e} ({-np.log2(e.max()):.1f} bits) rms {np.sqrt((e**2).mean()):.3e} ({-np.log2(np.sqrt((e**2).mean())):.1f} bits)")
# coefficient-domain analysis: decrypt input at 1 prime and output, compare polynomials
rin = vm.ciphers[0]; rout = vm.ciphers[vm.prog.res_dst[0]]
def coeffs(ct):
    pt = o.decrypt(ct); L = pt.ell
    c = o.ntt_inv(pt.data, list(range(L)))
    # centred CRT lift via python ints (small N only)
    Q = 1
    for q in o.primes[:L]: Q *= q
    out = np.zeros(o.N)
    res = [c[i].astype(object) for i in range(L)]
    x = np.zeros(o.N, dtype=object)
    for i,q in enumerate(o.primes[:L]):
        Qi = Q//q; x = (x + res[i]*(Qi*pow(Qi,-1,q))) % Q
    x = np.array([int(v) - Q if int(v) > Q//2 else int(v) for v in x], dtype=object)
    return np.array([float(v) for v in x]) / ct.scale
ci, co = coeffs(rin), coeffs(rout)
# I of the modraise: need <c1,s> stuff; instead look at the error by coefficient
ec = co - ci
print("coef err: max %.3e rms %.3e ; kurtosis %.1f" % (np.abs(ec).max(), np.sqrt((ec**2).mean()), ((ec**4).mean()/((ec**2).mean())**2)))
idx = np.argsort(-np.abs(ec))[:8]
print("worst coefs", idx, ec[idx], "msg coef there", ci[idx])
mr = o.modraise(rin, 3)
ct3 = mr
t = coeffs(type(rin)(mr.data, 1.0))
p1 = coeffs(type(rin)(rin.data, 1.0))
I = np.round((t - p1)/float(o.primes[0]))
print("I range", I.min(), I.max(), "std", I.std())
for v in range(int(I.min()), int(I.max())+1):
    m = I==v
    if m.sum(): print("  I=%3d n=%5d rms err %.3e max %.3e" % (v, m.sum(), np.sqrt((ec[m]**2).mean()), np.abs(ec[m]).max()))
