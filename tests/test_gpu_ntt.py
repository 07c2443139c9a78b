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


@pytest.mark.parametrize("logN,K,limbs,prime_major", [(15, 14, 520, 16), (17, 3, 136, 16), (17, 3, 135, 16), (17, 3, 135, 0), (16, 4, 160, 16)])
def test_large_batch_uses_the_throughput_tiles_and_matches_oracle(logN, K, limbs, prime_major, request):
    """>= 5000 workgroups per launch: the radix-8 (2048-coefficient) tile geometry (round 3: at N = 2^15 the roofline leg's forward
    transform takes the single-crossing kernel from 640 limbs on; this batch of 520 stays on the tiles, and N = 2^17 always does).
    Forward == oracle on a spread of limbs, repeated launches agree, forward/inverse round trips are exact.
    Round 5 (option rows_prime_major): from N = 2^16 a ROWS phase over whole periods of the prime pattern (135 = 45 x 3, 160 = 40 x 4) walks its
    limbs prime by prime -- the same transforms in another launch order; 136 limbs (not a multiple of the period) and option 0 keep limb order."""
    from dacapo_amd import runner

    runner.set_option("rows_prime_major", prime_major)
    request.addfinalizer(lambda: runner.set_option("rows_prime_major", 16))
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


@pytest.mark.parametrize("pairs", [(1, 1), (0, 0), (1, 0), (0, 1)])   # (ntt_full_pairs, ntt_full_inv_pairs): word and pair forms of either direction
def test_single_crossing_kernel_matches_oracle_and_the_two_launch_tiles(pairs):
    from dacapo_amd import runner

    with runner.options(ntt_full_pairs=pairs[0], ntt_full_inv_pairs=pairs[1]):
        _single_crossing_body()


def _single_crossing_body():
    """ntt_full.hip (one 1024-thread workgroup per limb of N = 2^15, one HBM crossing) through dc_ntt_variant: forward and inverse ==
    oracle bit for bit on an irregular prime pattern and on the edge limbs (zeros, q - 1), with a limb stride, and == the two-launch
    transform on a batch large enough that dc_ntt_forward itself takes the single-crossing kernel (>= 640 limbs)."""
    ll, ctx = _ctx(15, 14)
    o = Oracle(15, 14)
    N, K = 1 << 15, 14
    pidx = list(range(K)) + [K - 1, 0, 5]
    a = np.stack([splitmix_fill(0x51C0 + b, N) % np.uint64(o.primes[p]) for b, p in enumerate(pidx)])
    a[0, : N // 2] = 0
    a[1, :] = np.uint64(o.primes[pidx[1]] - 1)
    di = ll.DeviceBuffer.from_host(np.array(pidx, dtype=np.int32))
    d = ll.DeviceBuffer.from_host(a)
    ctx.ntt(d, len(pidx), prime_idx=di, variant=1)
    assert (d.to_host() == o.ntt_fwd(a, pidx)).all()
    ctx.ntt(d, len(pidx), inverse=True, prime_idx=di, variant=1)
    assert (d.to_host() == a).all()
    d2 = ll.DeviceBuffer.from_host(a)
    ctx.ntt(d2, len(pidx), inverse=True, prime_idx=di, variant=1)
    assert (d2.to_host() == o.ntt_inv(a, pidx)).all()
    # stride 2N, arithmetic prime pattern: the unused halves stay untouched
    s = np.zeros((6, 2, N), dtype=np.uint64)
    for b in range(6):
        s[b, 0] = splitmix_fill(b + 11, N) % np.uint64(o.primes[2 + b % 4])
        s[b, 1] = np.uint64(0xDEADBEEF)
    ds = ll.DeviceBuffer.from_host(s)
    ctx.ntt(ds, 6, prime_base=2, prime_period=4, limb_stride=2 * N, variant=1)
    got = ds.to_host()
    assert (got[:, 1] == np.uint64(0xDEADBEEF)).all()
    assert (got[:, 0] == o.ntt_fwd(s[:, 0], [2 + b % 4 for b in range(6)])).all()
    # a batch at which the library's own choice is the single-crossing kernel forward (>= 640 limbs; the inverse switches at 2048, and
    # variant=1 runs it here): more limbs than CUs, not a multiple of them, so the persistent grid's workgroups walk 4 or 5 limbs each
    limbs = 1040
    big = np.stack([splitmix_fill(0x4845564D + b, N) % np.uint64(o.primes[b % K]) for b in range(limbs)])
    outs = []
    for variant in (0, 1, None):
        db = ll.DeviceBuffer.from_host(big)
        ctx.ntt(db, limbs, prime_base=0, prime_period=K, variant=variant)
        outs.append(db.to_host())
        ctx.ntt(db, limbs, inverse=True, prime_base=0, prime_period=K, variant=variant)
        assert (db.to_host() == big).all()
    assert (outs[0] == outs[1]).all() and (outs[0] == outs[2]).all()
    check = [0, 1, 13, 14, 517, limbs - 1]
    assert (outs[1][check] == o.ntt_fwd(big[check], [b % K for b in check])).all()
