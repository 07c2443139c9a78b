"""GPU parity: HIP NTT / inverse NTT through the C ABI (include/dacapo_ckks.h) == CPU oracle, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.oracle import Oracle, splitmix_fill


def _ctx(logN, K):
    from dacapo_amd import lowlevel as ll

    return ll, ll.Context(logN, K)


@pytest.mark.parametrize("logN,K", [(12, 3), (13, 3), (14, 3), (15, 14), (16, 4), (17, 3)])
def test_ntt_matches_oracle(logN, K):
    ll, ctx = _ctx(logN, K)
    o = Oracle(logN, K)
    assert ctx.primes == o.primes
    assert ctx.roots == [o.psi(p) for p in range(K)]
    N = 1 << logN
    pidx = list(range(K)) + [K - 1, 0]  # an irregular prime pattern through the device index array
    a = np.stack([splitmix_fill(0x4845564D + b, N) % np.uint64(o.primes[p]) for b, p in enumerate(pidx)])
    # edge limbs: zeros / q-1 everywhere
    a[0, : N // 2] = 0
    a[1, :] = np.uint64(o.primes[pidx[1]] - 1)
    d = ll.DeviceBuffer.from_host(a)
    di = ll.DeviceBuffer.from_host(np.array(pidx, dtype=np.int32))
    ctx.ntt(d, len(pidx), prime_idx=di)
    got = d.to_host()
    want = o.ntt_fwd(a, pidx)
    assert (got == want).all()
    ctx.ntt(d, len(pidx), inverse=True, prime_idx=di)
    assert (d.to_host() == a).all()
    # inverse on fresh NTT-domain data == oracle inverse
    d2 = ll.DeviceBuffer.from_host(a)
    ctx.ntt(d2, len(pidx), inverse=True, prime_idx=di)
    assert (d2.to_host() == o.ntt_inv(a, pidx)).all()


def test_ntt_arithmetic_prime_pattern_and_stride():
    ll, ctx = _ctx(13, 4)
    o = Oracle(13, 4)
    N = 1 << 13
    # 6 limbs, stride 2N (every other slot unused), primes 1 + (b % 3)
    a = np.zeros((6, 2, N), dtype=np.uint64)
    for b in range(6):
        a[b, 0] = splitmix_fill(b + 1, N) % np.uint64(o.primes[1 + b % 3])
        a[b, 1] = np.uint64(0xDEADBEEF)
    d = ll.DeviceBuffer.from_host(a)
    ctx.ntt(d, 6, prime_base=1, prime_period=3, limb_stride=2 * N)
    got = d.to_host()
    assert (got[:, 1] == np.uint64(0xDEADBEEF)).all()
    assert (got[:, 0] == o.ntt_fwd(a[:, 0], [1 + b % 3 for b in range(6)])).all()


def test_linearity_and_convolution_at_full_size():
    """Size-independent properties at the reference ring N = 2^15 (BASELINE config sizes)."""
    ll, ctx = _ctx(15, 14)
    N, K = 1 << 15, 14
    q = np.array(ctx.primes, dtype=np.uint64)[:, None]
    x = np.stack([splitmix_fill(100 + i, N) for i in range(K)]) % q
    y = np.stack([splitmix_fill(200 + i, N) for i in range(K)]) % q
    s = (x + y) % q
    dx, dy, ds = (ll.DeviceBuffer.from_host(v) for v in (x, y, s))
    for d in (dx, dy, ds):
        ctx.ntt(d, K)
    fx, fy, fs = dx.to_host(), dy.to_host(), ds.to_host()
    assert (fs == (fx + fy) % q).all()  # NTT(x+y) == NTT(x)+NTT(y)
    # X * x(X): multiplying by the monomial is a negacyclic shift
    mono = np.zeros((K, N), dtype=np.uint64)
    mono[:, 1] = 1
    dm = ll.DeviceBuffer.from_host(mono)
    ctx.ntt(dm, K)
    dp = ll.DeviceBuffer((K, N))
    ll.lib().dc_poly_mul(ctx.h, dp.ptr, dx.ptr, dm.ptr, K, None)
    ctx.ntt(dp, K, inverse=True)
    shifted = np.roll(x, 1, axis=1)
    shifted[:, 0] = (q[:, 0] - x[:, -1]) % q[:, 0]
    assert (dp.to_host() == shifted).all()


@pytest.mark.parametrize("logN,K,limbs", [(15, 14, 520), (17, 3, 136)])
def test_large_batch_uses_the_throughput_tiles_and_matches_oracle(logN, K, limbs):
    """>= 5000 workgroups per launch: the radix-8 (2048-coefficient) tile geometry, the one bench.py's roofline leg times.
    Forward == oracle on a spread of limbs, repeated launches agree, forward/inverse round trips are exact."""
    ll, ctx = _ctx(logN, K)
    o = Oracle(logN, K)
    N = 1 << logN
    assert (N >> 11) * limbs >= 5000
    pidx = [b % K for b in range(limbs)]
    a = np.stack([splitmix_fill(0x4845564D + b, N) % np.uint64(o.primes[p]) for b, p in enumerate(pidx)])
    d = ll.DeviceBuffer.from_host(a)
    check = sorted(set(list(range(0, limbs, limbs // 24)) + [limbs - 1, limbs - 2, 7, 8]))
    want = o.ntt_fwd(a[check], [pidx[b] for b in check])
    first = None
    for _ in range(2):
        ctx.ntt(d, limbs, prime_base=0, prime_period=K)
        got = d.to_host()
        assert (got[check] == want).all()
        first = got if first is None else first
        assert (got == first).all()
        ctx.ntt(d, limbs, inverse=True, prime_base=0, prime_period=K)
        assert (d.to_host() == a).all()
