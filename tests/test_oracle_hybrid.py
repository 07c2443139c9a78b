"""CPU: the grouped-digit hybrid key switch of the oracle (oracle/ckks_oracle.c orc_keyswitch_hybrid; an EXTENSION -- what the reference's
HEaaN runtime does inside its closed library, HEAAN_HEVM.cpp:124-141 / :386-399 -- not SEAL's scheme):
  * with one special prime and one prime per digit it IS SEAL's switch_key_inplace: identical key layout and identical limbs;
  * with real groups (ks special primes, digits of alpha primes, including a partial last digit and levels below the chain's top) a
    rotation / relinearisation still decrypts to the right message, with noise of the same order as SEAL's scheme;
  * its closed form: the output equals round(sum_g d~_g key_g / P) limb for limb, recomputed with Python integers on a small ring."""
import numpy as np
import pytest

from oracle.oracle import Plaintext, Ciphertext, Oracle


def test_degenerate_hybrid_is_seals_key_switch():
    logN, K = 10, 5
    a, b = Oracle(logN, K), Oracle(logN, K)
    b.set_hybrid(1, 1)
    b.ks, b.alpha = 2, 2          # force the hybrid code path in the wrapper while the C context stays at (1, 1)
    a.keygen(seed=7, galois_elts=[3])
    b.rng.value = 0
    b.sk, b.pk = a.sk, a.pk
    # same randomness -> same key
    a2 = Oracle(logN, K)
    a2.keygen(seed=7, galois_elts=[3])
    assert (a2.galois[3] == a.galois[3]).all()
    rng = np.random.default_rng(1)
    for ell in (1, 2, 4):
        q = np.array(a.primes[:ell], dtype=np.uint64)[:, None]
        target = rng.integers(0, 1 << 62, size=(ell, a.N), dtype=np.uint64) % q
        o0, o1 = np.zeros((ell, a.N), dtype=np.uint64), np.zeros((ell, a.N), dtype=np.uint64)
        h0, h1 = o0.copy(), o1.copy()
        a.keyswitch(target, a.galois[3], o0, o1)
        b.L.orc_keyswitch_hybrid(b.ctx, ell, target.ctypes.data_as(__import__("ctypes").c_void_p), a.galois[3].ctypes.data_as(__import__("ctypes").c_void_p),
                                 h0.ctypes.data_as(__import__("ctypes").c_void_p), h1.ctypes.data_as(__import__("ctypes").c_void_p))
        assert (o0 == h0).all() and (o1 == h1).all()


# alpha <= ks: P must cover a digit (P >= Q_g) or the switching noise is Q_g / P times too large
@pytest.mark.parametrize("K,ks,alpha", [(7, 2, 2), (8, 3, 3), (9, 3, 2)])
def test_grouped_digits_rotate_and_relinearise_correctly(K, ks, alpha):
    logN = 10
    o = Oracle(logN, K)
    o.set_hybrid(ks, alpha)
    assert o.max_level == K - ks and o.dnum == -(-(K - ks) // alpha)
    o.keygen(seed=11, galois_elts=[o.elt_from_step(1), o.elt_from_step(-3)])
    assert o.relin.shape == (o.dnum, 2, K, o.N)
    rng = np.random.default_rng(2)
    x, y = rng.uniform(-1, 1, o.slots), rng.uniform(-1, 1, o.slots)
    for ell in range(1, o.max_level + 1):     # every level, including partial last digits
        cx, cy = o.encrypt(o.encode(x, 2.0**40, ell)), o.encrypt(o.encode(y, 2.0**40, ell))
        r = o.decode(o.decrypt(o.rotate(cx, 1)))
        assert np.abs(r - np.roll(x, -1)).max() < 1e-6, ell
        r = o.decode(o.decrypt(o.rotate(cx, -3)))
        assert np.abs(r - np.roll(x, 3)).max() < 1e-6, ell
        if ell >= 2:
            m = o.decode(o.decrypt(o.mul_relin(cx, cy)))
            assert np.abs(m - x * y).max() < 1e-6, ell


def test_hybrid_key_switch_closed_form_on_a_small_ring():
    """out = round-ish(sum_g d~_g key_g / P): the oracle's limbs recomputed from the definition with Python integers (CRT lifts, exact
    division), independent of the C code's RNS arithmetic: d~_g = sum_i [x_i qhat_i^-1]_{q_i} qhat_i as an INTEGER, the accumulated
    polynomial product mod Q_ell P, r = [acc + floor(P/2)]_P through the same integer base conversion, out = (acc - (conv - floor(P/2))) / P."""
    logN, K, ks, alpha = 4, 6, 2, 2
    o = Oracle(logN, K)
    o.set_hybrid(ks, alpha)
    o.keygen(seed=5, galois_elts=[3])
    N, L = o.N, K - ks
    key = o.galois[3]
    rng = np.random.default_rng(3)
    for ell in (1, 2, 3, 4):
        q = o.primes[:ell]
        sp = o.primes[L:]
        P = 1
        for p in sp:
            P *= p
        target = rng.integers(0, 1 << 62, size=(ell, N), dtype=np.uint64) % np.array(q, dtype=np.uint64)[:, None]
        out0, out1 = np.zeros((ell, N), dtype=np.uint64), np.zeros((ell, N), dtype=np.uint64)
        o.keyswitch(target, key, out0, out1)
        coef = o.ntt_inv(target, list(range(ell)))
        mods = q + sp
        pidx = list(range(ell)) + list(range(L, K))
        # negacyclic product in the coefficient domain, per modulus, with Python ints
        def negacyclic(u, v, m):
            res = [0] * N
            for i in range(N):
                for j in range(N):
                    k = i + j
                    t = u[i] * v[j]
                    if k >= N:
                        res[k - N] = (res[k - N] - t) % m
                    else:
                        res[k] = (res[k] + t) % m
            return res
        acc = [[[0] * N for _ in mods] for _ in range(2)]
        G = -(-ell // alpha)
        for g in range(G):
            lo, hi = g * alpha, min((g + 1) * alpha, ell)
            Qg = 1
            for i in range(lo, hi):
                Qg *= q[i]
            dt = [0] * N
            for i in range(lo, hi):
                qh = Qg // q[i]
                inv = pow(qh, -1, q[i])
                for n in range(N):
                    dt[n] += (int(coef[i][n]) * inv % q[i]) * qh
            for kc in range(2):
                for mi, (m, pi) in enumerate(zip(mods, pidx)):
                    kcoef = [int(v) for v in o.ntt_inv(key[g, kc, pi : pi + 1], [pi])[0]]
                    prod = negacyclic([d % m for d in dt], kcoef, m)
                    acc[kc][mi] = [(a + b) % m for a, b in zip(acc[kc][mi], prod)]
        for kc, out in ((0, out0), (1, out1)):
            want = np.zeros((ell, N), dtype=np.uint64)
            for n in range(N):
                # r = [acc + floor(P/2)]_P via the integer fast base conversion from the special primes
                conv = 0
                for j, p in enumerate(sp):
                    ph = P // p
                    conv += ((acc[kc][ell + j][n] + (P // 2)) % p * pow(ph, -1, p) % p) * ph
                for i in range(ell):
                    t = (conv - P // 2) % q[i]
                    want[i, n] = (acc[kc][i][n] - t) * pow(P, -1, q[i]) % q[i]
            got = o.ntt_inv(out, list(range(ell)))
            assert (got == want).all(), (ell, kc)


@pytest.mark.parametrize("K,ks,alpha", [(7, 2, 2), (9, 3, 2)])
def test_lazy_sum_of_rotations_shares_one_mod_down(K, ks, alpha):
    """orc_rotate_acc_hybrid / orc_moddown_hybrid (the GPU VM's option hyb_lazy_sum): a "sum" of ONE rotation is the eager rotation limb
    for limb; a sum of three decrypts to the sum of the rotated messages and differs from the eager sum by the roundings only (at most the
    three eager roundings against the single lazy one: 2 units plus the conversions' overshoot, |d| <= 2 + 2 ks per limb)."""
    logN = 10
    o = Oracle(logN, K)
    o.set_hybrid(ks, alpha)
    steps = [1, -3, 8]
    o.keygen(seed=5, galois_elts=[o.elt_from_step(s) for s in steps])
    rng = np.random.default_rng(3)
    for ell in (o.max_level, 3):
        xs = [rng.uniform(-1, 1, o.slots) for _ in steps]
        cts = [o.encrypt(o.encode(x, 2.0**40, ell)) for x in xs]
        one = o.lazy_add(o.rotate_lazy(cts[0], steps[0], 0, 1), cts[1])
        ref = o.add(o.rotate(cts[0], steps[0]), cts[1])
        assert isinstance(one, Ciphertext) and (one.data == ref.data).all()
        lazy = o.rotate_lazy(cts[0], steps[0], 0, 3)
        lazy = o.lazy_add(lazy, cts[1])                                  # an ordinary term joins the base
        lazy = o.lazy_add(o.rotate_lazy(cts[1], steps[1], 0, 3), lazy)
        assert not isinstance(lazy, Ciphertext)
        lazy = o.lazy_add(lazy, o.rotate_lazy(cts[2], steps[2], 0, 3))
        assert isinstance(lazy, Ciphertext)
        eager = o.add(o.add(o.add(o.rotate(cts[0], steps[0]), cts[1]), o.rotate(cts[1], steps[1])), o.rotate(cts[2], steps[2]))
        want = sum(np.roll(x, -s) for x, s in zip(xs, steps)) + xs[1]
        assert np.abs(o.decode(o.decrypt(lazy)) - want).max() < 1e-6
        q = np.array(o.primes[:ell], dtype=np.int64)[None, :, None]
        diff = np.stack([o.ntt_inv(o.poly_sub(lazy.data[k], eager.data[k]), list(range(ell))) for k in range(2)])   # coefficient domain
        d = diff.astype(np.int64) % q
        d = np.where(d > q // 2, d - q, d)
        assert np.abs(d).max() <= 2 + 2 * ks and np.abs(d).max() > 0


@pytest.mark.parametrize("K,ks,alpha", [(7, 2, 2), (9, 3, 2)])
def test_lazy_sum_with_plaintext_products_in_the_raised_basis(K, ks, alpha):
    """Oracle.lazy_mul_plain (the GPU VM's option hyb_double_hoist): pt * rot(x) taken while the rotation's inner products are still in the
    raised basis -- galois(c0) * pt on the data primes, the accumulators * pt over the data AND the special primes -- then one mod-down for
    the sum.  A convolution sum_t pt_t * rot_t(x_t) + a bare rotation + an ordinary term decrypts to the cleartext value as well as the eager
    evaluation does, and is not the eager limbs (one rounding, taken after the plaintexts)."""
    logN = 10
    o = Oracle(logN, K)
    o.set_hybrid(ks, alpha)
    steps = [1, -3, 8]
    o.keygen(seed=6, galois_elts=[o.elt_from_step(s) for s in steps])
    rng = np.random.default_rng(8)
    for ell in (o.max_level, 3):
        xs = [rng.uniform(-1, 1, o.slots) for _ in range(3)]
        ws = [rng.uniform(-1, 1, o.slots) for _ in range(2)]
        cts = [o.encrypt(o.encode(x, 2.0**40, ell)) for x in xs]
        full = [o.encode(w, 2.0**40, K) for w in ws]                       # the same encoded polynomial over the whole chain ...
        pts = [Plaintext(f.data[:ell].copy(), f.scale) for f in full]      # ... its data-prime limbs, as mulcp reads them
        sps = [f.data[K - ks:].copy() for f in full]                       # ... and its special-prime limbs
        assert all((pt.data == o.encode(w, 2.0**40, ell).data).all() for pt, w in zip(pts, ws))
        bare = o.mul_plain(cts[2], o.encode(np.full(o.slots, 0.5), 2.0**40, ell))   # an ordinary term at the products' scale
        lazy = o.lazy_mul_plain(o.rotate_lazy(cts[0], steps[0], 0, 3), pts[0], sps[0])
        lazy = o.lazy_add(lazy, bare)
        lazy = o.lazy_add(lazy, o.lazy_mul_plain(o.rotate_lazy(cts[1], steps[1], 0, 3), pts[1], sps[1]))
        assert not isinstance(lazy, Ciphertext)
        lazy = o.lazy_add(lazy, o.rotate_lazy(bare, steps[2], 0, 3))       # a bare member at the same scale
        assert isinstance(lazy, Ciphertext) and lazy.scale == 2.0**80
        eager = o.add(o.add(o.add(o.mul_plain(o.rotate(cts[0], steps[0]), pts[0]), bare), o.mul_plain(o.rotate(cts[1], steps[1]), pts[1])),
                      o.rotate(bare, steps[2]))
        want = ws[0] * np.roll(xs[0], -steps[0]) + 0.5 * xs[2] + ws[1] * np.roll(xs[1], -steps[1]) + 0.5 * np.roll(xs[2], -steps[2])
        e_lazy, e_eager = np.abs(o.decode(o.decrypt(lazy)) - want).max(), np.abs(o.decode(o.decrypt(eager)) - want).max()
        assert e_lazy < 1e-6 and e_lazy < 4 * e_eager + 1e-9
        assert (lazy.data != eager.data).any()


def test_oracle_vm_replays_lazy_groups(tmp_path):
    """OracleVM.set_lazy_groups (what the GPU parity tests feed with hevm_plan_lazy_groups): a program Sum_g rot_g(inner_g) run eagerly and
    with its three rotations named as one group decrypts to the same values, the limbs differ (one rounding instead of three), a group of ONE
    rotation reproduces the eager limbs exactly, and a group that the program's dataflow contradicts -- one of its rotations is multiplied
    by a plaintext before anything adds it -- is reported, not silently computed."""
    from dacapo_amd import hevm_asm as ha
    from oracle.oracle import OracleVM

    logN, K, ks = 10, 7, 2
    slots = 1 << (logN - 1)
    rng = np.random.default_rng(4)
    b = ha.Builder(slots=slots, init_level=K - ks, policy="lazy", boot_level=K - ks, shadow=True)
    x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
    out = b.add(x, y)
    for g, k in enumerate((4, 8, 16)):
        out = b.add(out, b.rotate(b.add(b.mul_plain(x, [0.1 * (g + 1)]), b.mul_plain(y, [0.2])), k))
    bad = b.mul_plain(b.rotate(x, 2), [0.5])
    b.output(b.finish(out))
    b.output(b.finish(bad))
    cst, hv, _ = b.assemble()
    (tmp_path / "p.cst").write_bytes(cst)
    (tmp_path / "p.hevm").write_bytes(hv)
    o = Oracle(logN, K)
    o.set_hybrid(ks)
    o.keygen(seed=3)

    def run(groups):
        vm = OracleVM(o)
        vm.load(tmp_path / "p.cst", tmp_path / "p.hevm")
        vm.preprocess()
        o.rng.value = 99                                        # same encryption randomness in every run
        for i, a in enumerate(b.args):
            vm.encrypt(i, a.plain)
        if groups is not None:
            vm.set_lazy_groups(groups)
        vm.run()
        return vm

    ops = ha.unpack_hevm(hv)["ops"].tolist()
    rots = [i for i, op in enumerate(ops) if op[0] == ha.OP_ROTATE]
    assert len(rots) == 4                                       # the three giant steps, then the rotation by 2
    eager, lazy, single = run(None), run([rots[:3]]), run([[rots[0]]])
    r = eager.prog.res_dst[0]
    assert np.abs(lazy.decrypt(r) - b.expected()[0]).max() < 1e-6 and np.abs(eager.decrypt(r) - b.expected()[0]).max() < 1e-6
    assert (lazy.ciphers[r].data != eager.ciphers[r].data).any()
    assert (single.ciphers[r].data == eager.ciphers[r].data).all()
    with pytest.raises(RuntimeError, match="unfinished lazy sum"):
        run([[rots[2], rots[3]]])
