"""Integer model of modarith.hpp's mulmod_pair (the twiddle-pair multiply of the single-crossing NTT's forward passes A and B): the same five
32 x 32 + 64 multiply-adds and one fold, on Python integers with every 64-bit accumulator checked for overflow, over random and extreme
operands -- the range argument in the header's comment (columns below 2^64, the part above 2^60 in one word, product below 2q), executed.
The device code itself is checked bit for bit against the oracle by tests/test_gpu_ntt.py; this test pins the arithmetic it relies on."""
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
