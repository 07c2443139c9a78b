"""GPU parity: every evaluator call of the HEVM path (through the C ABI) == CPU oracle on the same limbs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.oracle import Ciphertext, Oracle, Plaintext, splitmix_fill


@pytest.fixture(scope="module")
def env():
    from dacapo_amd import lowlevel as ll

    logN, K = 13, 6
    o = Oracle(logN, K)
    o.keygen(seed=0x4845564D, galois_elts=[3, 2 * (1 << logN) - 1, pow(3, (1 << logN) // 2 - 4, 2 << logN)])
    ctx = ll.Context(logN, K)
    assert ctx.primes == o.primes
    return ll, ctx, o


def _rand_ct(o, ell, seed):
    q = np.array(o.primes[:ell], dtype=np.uint64)[:, None]
    return np.stack([np.stack([splitmix_fill(seed + 7 * p + i, o.N) for i in range(ell)]) % q for p in range(2)])


def _dev_ct(ll, ct, cap):
    """upload [2][ell][N] into a register of capacity `cap` limbs per poly (poly stride cap*N)"""
    buf = np.zeros((2, cap, ct.shape[2]), dtype=np.uint64)
    buf[:, : ct.shape[1]] = ct
    return ll.DeviceBuffer.from_host(buf)


@pytest.mark.parametrize("ell", [1, 2, 5])
def test_elementwise_ops(env, ell):
    ll, ctx, o = env
    L, N, cap = ll.lib(), o.N, o.K - 1
    a, b = _rand_ct(o, ell, 1), _rand_ct(o, ell, 2)
    a[0, 0, :4] = 0
    b[0, 0, :4] = 0
    a[1, 0, :4] = np.uint64(o.primes[0] - 1)
    b[1, 0, :4] = np.uint64(o.primes[0] - 1)
    pt = _rand_ct(o, ell, 3)[0]
    da, db, dpt = _dev_ct(ll, a, cap), _dev_ct(ll, b, cap), ll.DeviceBuffer.from_host(pt)
    dd = ll.DeviceBuffer((2, cap, N))
    st = cap * N
    A, B, P = Ciphertext(a, 1.0), Ciphertext(b, 1.0), Plaintext(pt, 1.0)
    L.dc_ct_add(ctx.h, dd.ptr, st, da.ptr, st, db.ptr, st, ell, None)
    assert (dd.to_host()[:, :ell] == o.add(A, B).data).all()
    L.dc_ct_negate(ctx.h, dd.ptr, st, da.ptr, st, ell, None)
    assert (dd.to_host()[:, :ell] == o.negate(A).data).all()
    L.dc_ct_add_plain(ctx.h, dd.ptr, st, da.ptr, st, dpt.ptr, ell, None)
    assert (dd.to_host()[:, :ell] == o.add_plain(A, P).data).all()
    L.dc_ct_mul_plain(ctx.h, dd.ptr, st, da.ptr, st, dpt.ptr, ell, None)
    assert (dd.to_host()[:, :ell] == o.mul_plain(A, P).data).all()
    # aliasing dst == lhs (ReuseBuffer does this)
    L.dc_ct_add(ctx.h, da.ptr, st, da.ptr, st, db.ptr, st, ell, None)
    assert (da.to_host()[:, :ell] == o.add(A, B).data).all()
    L.dc_ct_add_plain(ctx.h, db.ptr, st, db.ptr, st, dpt.ptr, ell, None)
    assert (db.to_host()[:, :ell] == o.add_plain(B, P).data).all()
    if ell > 1:
        L.dc_ct_modswitch(ctx.h, dd.ptr, st, da.ptr, st, ell, 1, None)
        assert (dd.to_host()[:, : ell - 1] == o.add(A, B).data[:, : ell - 1]).all()


def test_galois_permutation(env):
    ll, ctx, o = env
    L, N = ll.lib(), o.N
    ell = 3
    a = _rand_ct(o, ell, 5)
    da, dd = ll.DeviceBuffer.from_host(a), ll.DeviceBuffer((2, ell, N))
    for step in (1, -4, 0, 77):
        elt = o.elt_from_step(step)
        assert ctx.elt_from_step(step) == elt
        L.dc_galois_ntt(ctx.h, dd.ptr, ell * N, da.ptr, ell * N, elt, 2, ell, None)
        assert (dd.to_host() == o.galois_ntt(a, elt).reshape(2, ell, N)).all()
    assert ctx.elt_from_step(N // 2) == 0


@pytest.mark.parametrize("ell", [1, 2, 3, 5])
def test_keyswitch(env, ell):
    ll, ctx, o = env
    L, N = ll.lib(), o.N
    target = _rand_ct(o, ell, 11)[0]
    base = _rand_ct(o, ell, 12)
    want0, want1 = base[0].copy(), base[1].copy()
    o.keyswitch(target, o.relin, want0, want1)
    dk = ll.DeviceBuffer.from_host(o.relin)
    dt, dbase, dout = ll.DeviceBuffer.from_host(target), ll.DeviceBuffer.from_host(base), ll.DeviceBuffer((2, ell, N))
    L.dc_keyswitch(ctx.h, dout.ptr, ell * N, dbase.at(0), dbase.at(ell * N), dt.ptr, dk.ptr, ell, None)
    got = dout.to_host()
    assert (got[0] == want0).all() and (got[1] == want1).all()
    assert (dt.to_host() == target).all()  # target preserved
    # no base: pure switched pair
    z0, z1 = np.zeros_like(want0), np.zeros_like(want1)
    o.keyswitch(target, o.relin, z0, z1)
    L.dc_keyswitch(ctx.h, dout.ptr, ell * N, None, None, dt.ptr, dk.ptr, ell, None)
    got = dout.to_host()
    assert (got[0] == z0).all() and (got[1] == z1).all()


@pytest.mark.parametrize("ell", [2, 3, 5])
def test_rescale_mulrelin_rotate(env, ell):
    ll, ctx, o = env
    L, N, cap = ll.lib(), o.N, o.K - 1
    st = cap * N
    a, b = _rand_ct(o, ell, 21), _rand_ct(o, ell, 22)
    A, B = Ciphertext(a, 2.0**40), Ciphertext(b, 2.0**40)
    da, db, dd = _dev_ct(ll, a, cap), _dev_ct(ll, b, cap), ll.DeviceBuffer((2, cap, N))
    L.dc_ct_rescale(ctx.h, dd.ptr, st, da.ptr, st, ell, None)
    assert (dd.to_host()[:, : ell - 1] == o.rescale(A).data).all()
    drel = ll.DeviceBuffer.from_host(o.relin)
    L.dc_ct_mul_relin(ctx.h, dd.ptr, st, da.ptr, st, db.ptr, st, drel.ptr, ell, None)
    assert (dd.to_host()[:, :ell] == o.mul_relin(A, B).data).all()
    for elt, key in o.galois.items():
        dkey = ll.DeviceBuffer.from_host(key)
        L.dc_ct_rotate_hop(ctx.h, dd.ptr, st, da.ptr, st, elt, dkey.ptr, ell, None)
        assert (dd.to_host()[:, :ell] == o.apply_galois(A, elt).data).all()
    # in-place forms (dst aliases a source register)
    want = o.mul_relin(A, A).data
    L.dc_ct_mul_relin(ctx.h, da.ptr, st, da.ptr, st, da.ptr, st, drel.ptr, ell, None)
    assert (da.to_host()[:, :ell] == want).all()
    want = o.rescale(B).data
    L.dc_ct_rescale(ctx.h, db.ptr, st, db.ptr, st, ell, None)
    assert (db.to_host()[:, : ell - 1] == want).all()


def test_homomorphic_roundtrip_on_gpu(env):
    """encrypt (oracle) -> mul_relin + rescale + rotate on the GPU -> decrypt (oracle) ~= plaintext result."""
    ll, ctx, o = env
    L, N, cap = ll.lib(), o.N, o.K - 1
    st = cap * N
    rng = np.random.default_rng(3)
    x, y = rng.uniform(-1, 1, o.slots), rng.uniform(-1, 1, o.slots)
    cx, cy = o.encrypt(o.encode(x, 2.0**50, cap)), o.encrypt(o.encode(y, 2.0**50, cap))
    dx, dy = _dev_ct(ll, cx.data, cap), _dev_ct(ll, cy.data, cap)
    drel, dgal = ll.DeviceBuffer.from_host(o.relin), ll.DeviceBuffer.from_host(o.galois[3])
    L.dc_ct_mul_relin(ctx.h, dx.ptr, st, dx.ptr, st, dy.ptr, st, drel.ptr, cap, None)
    L.dc_ct_rescale(ctx.h, dx.ptr, st, dx.ptr, st, cap, None)
    L.dc_ct_rotate_hop(ctx.h, dy.ptr, st, dx.ptr, st, 3, dgal.ptr, cap - 1, None)
    res = Ciphertext(dy.to_host()[:, : cap - 1].copy(), 2.0**100 / o.primes[cap - 1])
    got = o.decode(o.decrypt(res))
    assert np.abs(got - np.roll(x * y, -1)).max() < 1e-5


def test_baseline_config3_mul_relin_n65536_l24():
    """BASELINE.json config 3: ct x ct multiply + relinearize at N = 2^16 with 24 data primes + 1 special prime
    (25 x 60-bit, the same CoeffModulus::Create rule).  SEAL itself cannot run this size; it is the same algorithm."""
    from dacapo_amd import lowlevel as ll

    logN, K = 16, 25
    o = Oracle(logN, K)
    o.keygen(seed=0x4845564D, galois_elts=[])
    ctx = ll.Context(logN, K)
    assert ctx.primes == o.primes
    L, N, ell = ll.lib(), o.N, K - 1
    a, b = _rand_ct(o, ell, 31), _rand_ct(o, ell, 32)
    da, db, dd = ll.DeviceBuffer.from_host(a), ll.DeviceBuffer.from_host(b), ll.DeviceBuffer((2, ell, N))
    drel = ll.DeviceBuffer.from_host(o.relin)
    st = ell * N
    L.dc_ct_mul_relin(ctx.h, dd.ptr, st, da.ptr, st, db.ptr, st, drel.ptr, ell, None)
    want = o.mul_relin(Ciphertext(a, 2.0**40), Ciphertext(b, 2.0**40)).data
    assert (dd.to_host() == want).all()
    # encrypt -> multiply -> decrypt round trip at this size
    rng = np.random.default_rng(5)
    x, y = rng.uniform(-1, 1, o.slots), rng.uniform(-1, 1, o.slots)
    cx, cy = o.encrypt(o.encode(x, 2.0**40, ell)), o.encrypt(o.encode(y, 2.0**40, ell))
    dx, dy = ll.DeviceBuffer.from_host(cx.data), ll.DeviceBuffer.from_host(cy.data)
    L.dc_ct_mul_relin(ctx.h, dd.ptr, st, dx.ptr, st, dy.ptr, st, drel.ptr, ell, None)
    got = o.decode(o.decrypt(Ciphertext(dd.to_host(), 2.0**80)))
    assert np.abs(got - x * y).max() < 1e-6


@pytest.mark.parametrize("opts", [{}, {"small_tile_wgs": 0, "wide_tile_wgs": 0}, {"cols_pairs": 0, "tiny_tile_wgs": 512}])
def test_reference_size_keyswitch_ops_l13(opts):
    """The reference ring and chain (N = 2^15, 14 x 60-bit, SEAL_HEVM.cpp:39-53) at the top level l = 13:
    rotate hop, mul+relin and rescale bit-exact vs the oracle (210 NTT-equivalents per key switch).  Second parameter set: the lift launch
    forced onto the radix-16 tiles that the 13-prime lowering's batches take by themselves (f_ks_lift_fcols_kernel<7, 4>); third: round 4's
    launch shapes (word twiddles in the COLS tiles, no one-butterfly tiles for the deep hop's small launches)."""
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    with runner.options(**opts):
        _keyswitch_ops_l13(ll)


def _keyswitch_ops_l13(ll):

    logN, K = 15, 14
    o = Oracle(logN, K)
    o.keygen(seed=0x4845564D, galois_elts=[3])
    ctx = ll.Context(logN, K)
    L, N, ell = ll.lib(), o.N, K - 1
    a, b = _rand_ct(o, ell, 41), _rand_ct(o, ell, 42)
    A, B = Ciphertext(a, 2.0**40), Ciphertext(b, 2.0**40)
    st = ell * N
    da, db, dd = ll.DeviceBuffer.from_host(a), ll.DeviceBuffer.from_host(b), ll.DeviceBuffer((2, ell, N))
    dgal, drel = ll.DeviceBuffer.from_host(o.galois[3]), ll.DeviceBuffer.from_host(o.relin)
    L.dc_ct_rotate_hop(ctx.h, dd.ptr, st, da.ptr, st, 3, dgal.ptr, ell, None)
    assert (dd.to_host() == o.apply_galois(A, 3).data).all()
    L.dc_ct_mul_relin(ctx.h, dd.ptr, st, da.ptr, st, db.ptr, st, drel.ptr, ell, None)
    want = o.mul_relin(A, B)
    assert (dd.to_host() == want.data).all()
    L.dc_ct_rescale(ctx.h, dd.ptr, st, dd.ptr, st, ell, None)
    assert (dd.to_host()[:, : ell - 1] == o.rescale(want).data).all()
