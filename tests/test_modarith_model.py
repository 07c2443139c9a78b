"""Integer models of the device arithmetic in dacapo_amd/csrc/modarith.hpp, on Python integers with every 64-bit accumulator checked for
overflow, over random and extreme operands -- the range arguments of the header's comments, executed:
  * mulmod_lazy / reduce_words / fold60 (the 60-bit chain and the narrower widths of the generic-width build) and the forward
    butterfly's fold schedule (ntt_tile.hpp: fold every third stage);
  * mulmod_pair, the twiddle-pair multiply of the single-crossing NTT's forward passes A and B (columns below 2^64, the part above 2^60
    in one word, product below 2q).
The device code itself is checked bit for bit against the oracle by the -m gpu tests; these tests pin the arithmetic it relies on."""
import random

import pytest

M32, M64 = (1 << 32) - 1, (1 << 64) - 1


def mad32(a, b, c):
    assert 0 <= a <= M32 and 0 <= b <= M32 and 0 <= c <= M64
    r = a * b + c
    assert r <= M64, "v_mad_u64_u32 would wrap"
    return r


def mulmod_pair(w, W, y, delta):
    y0, y1 = y & 0x7FFFFFFF, (y >> 31) & M32
    assert y >> 31 <= M32 >> 1, "operand must be below 2^62"
    c0 = mad32(W & M32, y1, mad32(w & M32, y0, 0))
    c1 = mad32(W >> 32, y1, mad32(w >> 32, y0, c0 >> 32))
    assert c1 < 1 << 60
    th = (c1 >> 28) & M32
    assert c1 >> 28 <= M32
    return mad32(th, delta, (c0 & M32) | ((c1 & 0x0FFFFFFF) << 32))


def chain_primes():
    """the reference's chain at N = 2^15 (SEAL CoeffModulus::Create: 60-bit primes = 1 mod 2N scanning down from 2^60) by Miller-Rabin"""
    def is_prime(n):
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if n % p == 0:
                return n == p
        d, s = n - 1, 0
        while d % 2 == 0:
            d, s = d // 2, s + 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(s - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True
    out, q = [], (1 << 60) + 1 - (1 << 16)
    while len(out) < 14:
        if is_prime(q):
            out.append(q)
        q -= 1 << 16
    return out


@pytest.mark.parametrize("q", chain_primes()[:3] + [(1 << 60) - (1 << 28) + 1, (1 << 60) - 1])  # + the widest delta the context accepts, and delta = 1
def test_pair_multiply_ranges_and_congruence(q):
    delta = (1 << 60) - q
    assert 0 < delta < 1 << 28
    rng = random.Random(q)
    ws = [0, 1, q - 1, q >> 1, (1 << 32) - 1, 1 << 32, (1 << 59) + ((1 << 32) - 1)] + [rng.randrange(q) for _ in range(300)]
    ys = [0, 1, (1 << 31) - 1, 1 << 31, (1 << 62) - 1, (1 << 62) - (1 << 31), 4 * q - 1, q - 1, 2 * q] + [rng.randrange(1 << 62) for _ in range(300)]
    for w in ws:
        W = (w << 31) % q
        for y in (ys if w in ws[:7] else ys[:9] + ys[9:40]):
            t = mulmod_pair(w, W, y, delta)
            assert t % q == w * y % q
            assert t < 2 * q  # what lets x' = fold(x) + t and y' = fold(x) + 2q - t stay below 4q < 2^62


def test_forward_butterfly_keeps_every_value_below_2_to_62():
    """ct_bfly_p: xf = fold(x) < 2q, t < 2q -> x' = xf + t < 4q, y' = xf + 2q - t in (0, 4q): the next stage's operand range"""
    q = chain_primes()[0]
    delta = (1 << 60) - q

    def fold60(x):
        return ((x >> 60) & M32) * delta + (x & ((1 << 60) - 1))

    rng = random.Random(7)
    for _ in range(2000):
        x, y, w = rng.randrange(4 * q), rng.randrange(4 * q), rng.randrange(q)
        xf, t = fold60(x), mulmod_pair(w, (w << 31) % q, y, delta)
        assert xf < 2 * q and xf % q == x % q
        xo, yo = xf + t, xf + 2 * q - t
        assert 0 < yo and xo < 4 * q and yo < 4 * q and 4 * q < 1 << 62
        assert xo % q == (x + w * y) % q and yo % q == (x - w * y) % q


# ---- mulmod_lazy / reduce_words / fold60 (modarith.hpp), for a prime q = 2^b - d, 45 <= b <= 60 ---------------------------------------------
def fold_b(x, b, d):
    assert 0 <= x <= M64
    return mad32((x >> 32) >> (b - 32), d, (x & M32) | (((x >> 32) & ((1 << (b - 32)) - 1)) << 32))


def reduce_words(hi, w1, w0, b, d):
    sh, mask = b - 32, (1 << (b - 32)) - 1
    assert hi < 1 << b, "precondition: T < 2^(2b+4)"
    H0 = (((hi & M32) << 32 | w1) >> sh) & M32
    H1 = (hi >> sh) & M32
    assert hi >> sh <= M32
    A = mad32(H0, d, w0 | ((w1 & mask) << 32))
    Bv = mad32(H1, d, 0)
    C = mad32((Bv >> sh) & M32, d, A)
    assert Bv >> sh <= M32
    hi_word = (C >> 32) + (Bv & mask)
    assert hi_word <= M32, "the 32-bit add on the high word would wrap"
    R = (C & M32) | (hi_word << 32)
    return fold_b(R, b, d) if b < 60 else R


def mulmod_lazy(a, y, b, d):
    a0, a1, b0, b1 = a & M32, a >> 32, y & M32, y >> 32
    p00 = mad32(a0, b0, 0)
    mid = mad32(a1, b0, mad32(a0, b1, p00 >> 32))
    hi = mad32(a1, b1, mid >> 32)
    return reduce_words(hi, mid & M32, p00 & M32, b, d)


@pytest.mark.parametrize("b,d", [(60, (1 << 60) - chain_primes()[0]), (60, (1 << 28) - 1), (60, 1), (51, (1 << 28) - 1), (51, (1 << 26) + 12345), (46, (1 << 27) + 99), (45, (1 << 26) - 1000)])
def test_lazy_multiply_ranges_and_congruence(b, d):
    q = (1 << b) - d  # (primality plays no part in the range argument)
    assert d < 1 << 28 and (b == 60 or (d << (64 - b)) < q)  # modarith.hpp prime_shape_ok: what the context accepts
    rng = random.Random(b * 1000003 + d)
    top = (15 << (b + 3)) // 8 - 1  # multiplicand bound of the fold schedule: 1.875 * 2^(b+3) (1.875 * 2^63 for the reference's chain)
    for a in [0, 1, q - 1, (1 << 32) - 1, 1 << 32] + [rng.randrange(q) for _ in range(400)]:
        for y in [0, 1, q - 1, 4 * q - 1, top, top - M32, (1 << (b + 2)) - 1] + [rng.randrange(top + 1) for _ in range(40)]:
            t = mulmod_lazy(a, y, b, d)
            assert t % q == a * y % q
            assert t < (1 << (b + 2)) and t < 4 * q + 4 * d  # "< 2^62" for b = 60; the narrow widths' third fold leaves < 2q
            if b < 60:
                assert t < 2 * q
    for x in [0, M64, 1 << 63, (1 << b) - 1, 1 << b] + [rng.randrange(1 << 64) for _ in range(2000)]:
        f = fold_b(x, b, d)
        assert f % q == x % q and f < 2 * q


def test_forward_fold_schedule_keeps_sums_in_64_bits():
    """ntt_tile.hpp: F N N F N N ... -- after an F stage every output is below 1.25 * 2^62 + 2^32, an N stage adds at most 2^62, the next
    multiplicand stays below 1.875 * 2^63; executed on the bounds themselves for the widest delta the context accepts"""
    b, d = 60, (1 << 28) - 1
    q = (1 << b) - d
    t_max = 4 * q + 4 * d                    # mulmod_lazy's output bound (< 2^62)
    bound = q                                # canonical input
    for stage in range(17):                  # the longest phase has 9 stages; two phases back to back cover every start
        folds = stage % 3 == 0
        xf = 2 * q if folds else bound       # fold60 output < 2q
        assert bound <= (15 << 63) // 8      # the multiplicand of THIS stage is a previous output
        bound = max(xf + t_max, xf + 4 * q)  # x' = xf + t ; y' = xf + 4q - t
        assert bound <= M64


def cols_tile_of(bx, K, LOGE, logN):
    """ntt_tile.hpp cols_tile_of, restated: workgroup index of a COLS launch -> tile of the limb"""
    log_tile = 8 + LOGE
    logb = log_tile - K                       # log2(columns of a tile)
    logs = 0 if logb >= 4 else 4 - logb       # log2(tiles that share a 128-byte line)
    if logs == 0 or logN - log_tile < 3 + logs:
        return bx
    span = 8 << logs
    r = bx & (span - 1)
    return (bx & ~(span - 1)) | ((r & 7) << logs) | (r >> 3)


@pytest.mark.parametrize("logN,K", [(15, 7), (16, 8), (17, 8), (14, 7), (13, 6)])
@pytest.mark.parametrize("LOGE", [1, 2, 3])
def test_cols_tile_placement_is_a_bijection_and_line_sharers_share_an_xcd(logN, K, LOGE):
    """ntt_tile.hpp cols_tile_of: a COLS tile is 2^(8 + LOGE - K) columns = that many 8-byte words of every row it touches; the tiles whose
    columns fall into one 128-byte line (16 words) must be handed to workgroups that are congruent mod 8 (workgroup w runs on XCD w mod 8:
    tools/experiments/xcc_probe.hip) and close in launch order, and the map must stay a permutation of the limb's tiles"""
    tiles = 1 << (logN - 8 - LOGE)
    cols = 1 << (8 + LOGE - K)
    img = [cols_tile_of(bx, K, LOGE, logN) for bx in range(tiles)]
    assert sorted(img) == list(range(tiles))
    sharers = max(1, 16 // cols)
    if sharers > 1 and tiles >= 8 * sharers:
        wg_of = {t: bx for bx, t in enumerate(img)}
        for line in range(tiles // sharers):
            wgs = [wg_of[line * sharers + j] for j in range(sharers)]
            assert len({w % 8 for w in wgs}) == 1, (line, wgs)            # one XCD
            assert max(wgs) - min(wgs) == 8 * (sharers - 1), (line, wgs)  # dispatched within 8 * sharers workgroups of each other
    else:
        assert img == list(range(tiles))
