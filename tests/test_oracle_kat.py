"""Pins the CPU oracle (oracle/) -- number theory, tables, NTT -- against SEAL known answers and definitions.

No GPU. The reference (/root/reference) ships no golden vectors for this path; the SEAL
constants below are the known-answer values of SEAL 4.0's own unit test tests/seal/util/ntt.cpp
(NTTTablesTest.NTTBasics / NTTPrimitiveRootsTest) and of its 60-bit prime table, recalled from upstream.
"""
import numpy as np
import pytest

from oracle.oracle import Oracle, lib, splitmix_fill

SEAL_Q60 = 0xFFFFFFFFFFC0001  # first 60-bit prime SEAL finds for 2N = 2^16 (and its classic test modulus)

# CoeffModulus::Create(32768, {60 x 14}) as used by SEAL_HEVM.cpp:48-53 (computed independently)
HEVM_CHAIN = [
    0xFFFFFFFFE7C0001, 0xFFFFFFFFE830001, 0xFFFFFFFFE9E0001, 0xFFFFFFFFEBB0001, 0xFFFFFFFFECA0001,
    0xFFFFFFFFEFE0001, 0xFFFFFFFFF240001, 0xFFFFFFFFF2A0001, 0xFFFFFFFFF330001, 0xFFFFFFFFF550001,
    0xFFFFFFFFF5A0001, 0xFFFFFFFFF6A0001, 0xFFFFFFFFF840001, 0xFFFFFFFFFFC0001,
]


def test_prime_chain_matches_seal_create():
    o = Oracle(15, 14)
    assert o.primes == HEVM_CHAIN
    assert o.primes[-1] == SEAL_Q60  # special prime = first prime found scanning down from 2^60
    for q in o.primes:
        assert q % (2 << 15) == 1 and q.bit_length() == 60
        assert (1 << 60) - q < (1 << 32)  # the 2^60 - delta shape the HIP reduction relies on
    # every candidate skipped between consecutive chain members is composite (nothing was missed)
    L = lib()
    v, found = ((1 << 60) - 1) // (1 << 16) * (1 << 16) + 1, []
    while len(found) < 14:
        if L.orc_is_prime(v):
            found.append(v)
        v -= 1 << 16
    assert found[::-1] == HEVM_CHAIN


def test_is_prime_against_trial_division():
    L = lib()
    for n in list(range(0, 2000)) + [2**31 - 1, 2**31 + 1, 2**61 - 1, 2**61 + 1, 3215031751, 18446744073709551557]:
        ref = n >= 2 and all(n % d for d in range(2, int(n**0.5) + 1)) if n < 10**7 else None
        if ref is not None:
            assert bool(L.orc_is_prime(n)) == ref, n
    assert L.orc_is_prime(2**61 - 1) and not L.orc_is_prime(2**61 + 1)
    assert not L.orc_is_prime(3215031751)  # strong pseudoprime to bases 2,3,5,7
    assert L.orc_is_prime(18446744073709551557)  # largest 64-bit prime


def test_seal_ntt_tables_known_answers():
    """SEAL tests/seal/util/ntt.cpp: NTTTables(coeff_count_power, Modulus(0xffffffffffc0001))."""
    L = lib()
    assert L.orc_min_primitive_root(4, SEAL_Q60) == 288794978602139552
    t1 = Oracle(1, 1, primes=[SEAL_Q60]).root_powers(0)
    assert [int(x) for x in t1] == [1, 288794978602139552]
    t2 = Oracle(2, 1, primes=[SEAL_Q60]).root_powers(0)
    assert [int(x) for x in t2] == [1, 288794978602139552, 178930308976060547, 748001537669050592]
    # algebraic re-verification of the recalled constants
    psi = 178930308976060547
    assert pow(psi, 4, SEAL_Q60) == SEAL_Q60 - 1 and pow(psi, 2, SEAL_Q60) == 288794978602139552
    assert min(pow(psi, e, SEAL_Q60) for e in (1, 3, 5, 7)) == psi


@pytest.mark.parametrize("logn", [3, 6, 10])
def test_root_tables(logn):
    o = Oracle(logn, 3)
    N = o.N
    for p, q in enumerate(o.primes):
        psi = o.psi(p)
        assert pow(psi, N, q) == q - 1
        # minimal among all primitive 2N-th roots
        if N <= 64:
            assert psi == min(pow(psi, e, q) for e in range(1, 2 * N, 2))
        rp, irp = o.root_powers(p), o.inv_root_powers(p)
        for k in (0, 1, 2, 3, N // 2, N - 1):
            br = int(f"{k:0{logn}b}"[::-1], 2)
            assert int(rp[k]) == pow(psi, br, q)
            assert int(rp[k]) * int(irp[k]) % q == 1


@pytest.mark.parametrize("logn", [1, 2, 5, 8])
def test_ntt_equals_definition(logn):
    o = Oracle(logn, 2)
    for p, q in enumerate(o.primes):
        a = splitmix_fill(7 + p, o.N) % np.uint64(q)
        want = o.ntt_fwd_definition(a, p)
        assert (o.ntt_fwd(a[None], [p])[0] == want).all()
        assert (o.ntt_fwd_simple(a, p) == want).all()
        assert (o.ntt_inv(want[None], [p])[0] == a).all()
        assert (o.ntt_inv_simple(want, p) == a).all()


def test_ntt_edge_values(oracle_small):
    o = oracle_small
    for p, q in enumerate(o.primes):
        for fill in (0, 1, q - 1):
            a = np.full(o.N, fill, dtype=np.uint64)
            f = o.ntt_fwd(a[None], [p])[0]
            assert (f == o.ntt_fwd_simple(a, p)).all() and (f < q).all()
            assert (o.ntt_inv(f[None], [p])[0] == a).all()
        # delta at X^0 -> all ones ; X^1 -> odd powers of psi in bit-reversed order
        d = np.zeros(o.N, dtype=np.uint64)
        d[0] = 1
        assert (o.ntt_fwd(d[None], [p]) == 1).all()


def test_dyadic_product_is_negacyclic_convolution(oracle_small):
    o = oracle_small
    for p, q in enumerate(o.primes[:2]):
        a = splitmix_fill(11, o.N) % np.uint64(q)
        b = splitmix_fill(12, o.N) % np.uint64(q)
        fa, fb = o.ntt_fwd(a[None], [p])[0], o.ntt_fwd(b[None], [p])[0]
        prod = np.array([int(x) * int(y) % q for x, y in zip(fa, fb)], dtype=np.uint64)
        assert (o.ntt_inv(prod[None], [p])[0] == o.negacyclic_schoolbook(a, b, p)).all()


def test_barrett_matches_percent(oracle_small):
    o = oracle_small
    ell = 4
    a = np.stack([splitmix_fill(20 + i, o.N) % np.uint64(q) for i, q in enumerate(o.primes[:ell])])
    b = np.stack([splitmix_fill(30 + i, o.N) % np.uint64(q) for i, q in enumerate(o.primes[:ell])])
    a[:, 0], b[:, 0] = 0, 0
    a[:, 1] = b[:, 1] = np.array(o.primes[:ell], dtype=np.uint64) - np.uint64(1)
    assert (o.poly_mul(a, b) == o.poly_mul_simple(a, b)).all()
    s = o.poly_add(a, b)
    for i, q in enumerate(o.primes[:ell]):
        assert [int(x) for x in s[i][:8]] == [(int(x) + int(y)) % q for x, y in zip(a[i][:8], b[i][:8])]
        assert (o.poly_add(o.poly_neg(a), a)[i] == 0).all()
        assert (o.poly_sub(a, b)[i] == o.poly_add(a, o.poly_neg(b))[i]).all()
