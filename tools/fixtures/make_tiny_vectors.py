#!/usr/bin/env python3
"""Hand-auditable known-answer vectors for rescale, one Galois key-switch hop and ct x ct + relinearise on a tiny ring
(N = 8, three 60-bit primes + one special prime), computed ONLY from the mathematical definitions with Python integers:

  * no NTT algorithm: "NTT form" is evaluation at the odd powers of psi, out[i] = a(psi^(2 brev(i) + 1)) (Horner, O(N^2)),
    psi = the numerically smallest primitive 2N-th root of unity mod q, found by exhaustive search
    [SEAL-upstream util/ntt.cpp NTTTables::initialize, numth.cpp try_minimal_primitive_root];
  * no RNS arithmetic: every polynomial is lifted to its canonical CRT representative over Z (coefficients in [0, Q)) and the
    operation is its closed form over the integers:
        rescale               out = floor((x + floor(p/2)) / p) mod q_i,        p = the dropped prime, x in [0, Q)
        key-switch mod-down   out = floor((y + floor(P/2)) / P) mod q_i,        y in [0, Q P) the lift of  sum_j d_j * key_j
        digit d_j             = the target's coefficient-domain residue mod q_j in [0, q_j), read as an integer
        Galois                c(X) -> c(X^e)  on coefficients (X^N = -1)
        multiply              (a0 b0, a0 b1 + a1 b0) + key-switch(a1 b1)        (negacyclic products over Z, reduced mod q_i)
    which is what Evaluator::rescale_to_next / apply_galois_inplace / multiply + relinearize_inplace of SEAL 4.0 compute
    (the calls of /root/reference/lib/Runtime/SEAL_HEVM.cpp:283, :273, :315-316) [SEAL-upstream evaluator.cpp, rns.cpp];
  * primes as CoeffModulus::Create(N, {60,60,60,60}) picks them: scan down from 2^60 in steps of 2N, keep primes; the first
    found is the special prime (last in the chain), the last found is q_0.

Inputs are produced by a 3-line LCG so that anyone can regenerate them.  The committed output, tests/golden/tiny_vectors.json,
pins oracle/ckks_oracle.c (tests/test_tiny_vectors.py); nothing here imports the oracle or the product.

    python tools/fixtures/make_tiny_vectors.py > tests/golden/tiny_vectors.json
"""
import json

N, LOGN = 8, 3


def is_prime(n):  # deterministic Miller-Rabin for n < 2^64
    if n < 2:
        return False
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


def seal_primes(count):
    found, v = [], (1 << 60) - 2 * N + 1
    while len(found) < count:
        if is_prime(v):
            found.append(v)
        v -= 2 * N
    return found[::-1]  # last found first; the first found (largest) is the special prime


def min_primitive_root(q):
    """smallest psi with psi^N = -1 mod q (i.e. of exact order 2N)"""
    g = next(g for g in range(2, 1000) if pow(g, (q - 1) // 2, q) == q - 1)  # a non-residue: its (q-1)/2N-th power has order 2N
    r = pow(g, (q - 1) // (2 * N), q)
    return min(pow(r, k, q) for k in range(1, 2 * N, 2))  # all primitive 2N-th roots are the odd powers of one of them


def brev(i):
    return int(format(i, f"0{LOGN}b")[::-1], 2)


def to_ntt(a, q, psi):
    """coefficients -> SEAL's bit-reversed evaluation order"""
    out = []
    for i in range(N):
        x, acc = pow(psi, 2 * brev(i) + 1, q), 0
        for c in reversed(a):
            acc = (acc * x + c) % q
        out.append(acc)
    return out


def crt_lift(residues, moduli):
    """the integer in [0, prod moduli) with the given residues"""
    x, m = 0, 1
    for r, q in zip(residues, moduli):
        x += m * (((r - x) * pow(m, -1, q)) % q)
        m *= q
    return x


def negacyclic_mul(a, b):
    """over Z, modulo X^N + 1"""
    out = [0] * N
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            k = i + j
            out[k % N] += x * y if k < N else -x * y
    return out


def galois_coeff(a, e):
    """a(X) -> a(X^e) mod X^N + 1, integer coefficients"""
    out = [0] * N
    for i, c in enumerate(a):
        k = i * e % (2 * N)
        out[k % N] += c if k < N else -c
    return out


class Lcg:
    def __init__(self, seed):
        self.s = seed

    def next(self):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return self.s >> 4  # 60 bits


def main():
    primes = seal_primes(4)                  # q0, q1, q2, P
    q, P = primes[:3], primes[3]
    psi = [min_primitive_root(p) for p in primes]
    rnd = Lcg(2026)
    ell = 3
    small = lambda: [rnd.next() % 3 - 1 for _ in range(N)]        # noqa: E731  ternary
    noise = lambda: [rnd.next() % 7 - 3 for _ in range(N)]        # noqa: E731  small error
    uniform = lambda mods: [[rnd.next() % m for _ in range(N)] for m in mods]  # noqa: E731  one residue polynomial per modulus

    s = small()                                                   # secret key, integer coefficients in {-1, 0, 1}
    e_gal = 3                                                     # Galois element of "rotate left by 1"

    def kswitch_key(new_key):
        """digit j: (-(a_j s + e_j) + [only limb j] (P mod q_j) new_key, a_j) over {q0, q1, q2, P}, coefficient domain residues
        [SEAL-upstream KeyGenerator::generate_one_kswitch_key]"""
        key = []
        for j in range(ell):
            a = uniform(primes)
            e = noise()
            c0 = []
            for i, m in enumerate(primes):
                t = negacyclic_mul(a[i], s)
                row = [(-(t[k] + e[k])) % m for k in range(N)]
                if i == j:
                    row = [(row[k] + (P % m) * new_key[k]) % m for k in range(N)]
                c0.append(row)
            key.append((c0, a))
        return key

    def key_switch(target_coeff, key):
        """target: residues [ell][N] in the coefficient domain -> (ks0, ks1) residues mod q_i, by the closed form"""
        out = []
        for part in (0, 1):
            acc = [[0] * N for _ in primes]                       # sum_j d_j * key_j[part] over {q_i, P}
            for j in range(ell):
                d = target_coeff[j]                               # integers in [0, q_j)
                for i, m in enumerate(primes):
                    t = negacyclic_mul(d, key[j][part][i])
                    acc[i] = [(acc[i][k] + t[k]) % m for k in range(N)]
            rows = [[0] * N for _ in q]
            for k in range(N):
                y = crt_lift([acc[i][k] for i in range(4)], primes)
                r = (y + P // 2) // P
                for i in range(ell):
                    rows[i][k] = r % q[i]
            out.append(rows)
        return out

    ct = [uniform(q), uniform(q)]                                 # a "ciphertext": two uniformly random polynomials (residues mod q_i)
    ct_b = [uniform(q), uniform(q)]

    # ---- rescale: drop q2 --------------------------------------------------------------------------------------------------
    rescaled = []
    for poly in ct:
        rows = [[0] * N for _ in range(2)]
        for k in range(N):
            x = crt_lift([poly[i][k] for i in range(3)], q)
            r = (x + q[2] // 2) // q[2]
            rows[0][k], rows[1][k] = r % q[0], r % q[1]
        rescaled.append(rows)

    # ---- one rotation hop by Galois element 3 --------------------------------------------------------------------------------
    gal_key = kswitch_key(galois_coeff(s, e_gal))
    rot_c0 = [[c % q[i] for c in galois_coeff(ct[0][i], e_gal)] for i in range(ell)]
    rot_c1 = [[c % q[i] for c in galois_coeff(ct[1][i], e_gal)] for i in range(ell)]
    ks = key_switch(rot_c1, gal_key)
    rotated = [[[(rot_c0[i][k] + ks[0][i][k]) % q[i] for k in range(N)] for i in range(ell)], ks[1]]

    # ---- multiply + relinearise -----------------------------------------------------------------------------------------------
    relin_key = kswitch_key(negacyclic_mul(s, s))
    prod = lambda x, y, i: [c % q[i] for c in negacyclic_mul(x[i], y[i])]  # noqa: E731
    d0 = [prod(ct[0], ct_b[0], i) for i in range(ell)]
    d1 = [[(u + v) % q[i] for u, v in zip(prod(ct[0], ct_b[1], i), prod(ct[1], ct_b[0], i))] for i in range(ell)]
    d2 = [prod(ct[1], ct_b[1], i) for i in range(ell)]
    ks = key_switch(d2, relin_key)
    mulled = [[[(d0[i][k] + ks[0][i][k]) % q[i] for k in range(N)] for i in range(ell)],
              [[(d1[i][k] + ks[1][i][k]) % q[i] for k in range(N)] for i in range(ell)]]

    ntt_ct = lambda c, mods=q, roots=psi: [[to_ntt(poly[i], mods[i], roots[i]) for i in range(len(poly))] for poly in c]  # noqa: E731
    ntt_key = lambda key: [[[to_ntt(part[i], primes[i], psi[i]) for i in range(4)] for part in digit] for digit in key]  # noqa: E731
    out = {
        "about": "tools/fixtures/make_tiny_vectors.py: closed forms over the integers on Z[X]/(X^8+1); all polynomials below are in NTT form "
                 "(out[i] = a(psi^(2 brev(i)+1))), limb-major, as decimal strings",
        "logN": LOGN, "primes": primes, "psi": psi, "galois_elt": e_gal,
        "secret_key_coefficients": s,
        "ct_a": ntt_ct(ct), "ct_b": ntt_ct(ct_b),
        "galois_key": ntt_key(gal_key), "relin_key": ntt_key(relin_key),
        "expect_rescale_a": ntt_ct(rescaled), "expect_rotate_a": ntt_ct(rotated), "expect_mul_relin_ab": ntt_ct(mulled),
        "ntt_example": {"coefficients": ct[0][0], "evaluations": to_ntt(ct[0][0], q[0], psi[0]), "prime": q[0], "psi": psi[0]},
    }
    print(json.dumps(json.loads(json.dumps(out), parse_int=str), indent=1))


if __name__ == "__main__":
    main()
