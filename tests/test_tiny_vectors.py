"""CPU: the oracle against hand-auditable vectors (tests/golden/tiny_vectors.json, made by tools/fixtures/make_tiny_vectors.py from the
integer closed forms of rescale / key switch / multiply on Z[X]/(X^8+1) -- no NTT algorithm, no RNS tricks, no oracle code).
What SEAL_HEVM.cpp:283 (rescale_to_next), :273 (rotate_vector -> apply_galois + switch_key) and :315-316 (multiply +
relinearize) compute, small enough to recompute by hand or with any big-integer calculator."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

from oracle.oracle import Ciphertext, Oracle

ROOT = Path(__file__).resolve().parent.parent
V = json.loads((ROOT / "tests" / "golden" / "tiny_vectors.json").read_text())


def _ints(x):
    """the JSON holds every integer as a decimal string (60-bit values do not survive a float-based JSON reader)"""
    return int(x) if isinstance(x, str) else [_ints(y) for y in x]


def to_u64(nested):
    return np.array(_ints(nested), dtype=np.uint64)


def test_committed_vectors_are_what_the_generator_prints():
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "fixtures" / "make_tiny_vectors.py")], capture_output=True, text=True, check=True).stdout
    assert json.loads(out) == V


def test_chain_and_roots_match_the_definitions():
    o = Oracle(int(V["logN"]), 4)
    assert o.primes == _ints(V["primes"])           # CoeffModulus::Create order: last found first, special prime last
    assert [o.psi(i) for i in range(4)] == _ints(V["psi"])  # numerically smallest primitive 2N-th roots
    ex = V["ntt_example"]
    a = to_u64([ex["coefficients"]])
    assert o.ntt_fwd(a, [0])[0].tolist() == _ints(ex["evaluations"])  # bit-reversed evaluation order


def test_rescale_rotate_and_multiply_match_the_integer_closed_forms():
    o = Oracle(int(V["logN"]), 4)
    A, B = Ciphertext(to_u64(V["ct_a"]), 2.0**40), Ciphertext(to_u64(V["ct_b"]), 2.0**40)
    assert A.data.shape == (2, 3, 8)
    got = o.rescale(A)
    assert got.data.tolist() == _ints(V["expect_rescale_a"])
    assert got.scale == 2.0**40 / float(o.primes[2])
    elt = int(V["galois_elt"])
    o.galois = {elt: to_u64(V["galois_key"])}
    assert o.galois[elt].shape == (3, 2, 4, 8)
    assert o.apply_galois(A, elt).data.tolist() == _ints(V["expect_rotate_a"])
    o.relin = to_u64(V["relin_key"])
    assert o.mul_relin(A, B).data.tolist() == _ints(V["expect_mul_relin_ab"])


def test_the_vectors_decrypt_consistently():
    """sanity of the vectors themselves under the oracle's arithmetic: (rotated c0 + c1 s) = Galois image of (c0 + c1 s) up to
    key-switch noise -- the keys in the file are real RLWE key-switch keys for the committed secret"""
    o = Oracle(int(V["logN"]), 4)
    s = np.array([int(x) for x in V["secret_key_coefficients"]], dtype=np.int64)
    q = o.primes
    sk = np.stack([np.where(s < 0, np.int64(0), s).astype(np.uint64) + np.where(s < 0, np.uint64(q[i] - 1), np.uint64(0)) for i in range(4)])
    sk = o.ntt_fwd(sk, [0, 1, 2, 3])
    A = Ciphertext(to_u64(V["ct_a"]), 1.0)
    R = Ciphertext(to_u64(V["expect_rotate_a"]), 1.0)
    dec = lambda c: o.poly_add(c.data[0], o.poly_mul(c.data[1], sk[:3]))  # noqa: E731
    want = o.galois_ntt(dec(A), int(V["galois_elt"]))
    diff = o.ntt_inv(o.poly_sub(dec(R), want), [0, 1, 2])[0].astype(object)
    diff = [int(d) if d < q[0] // 2 else int(d) - q[0] for d in diff]
    assert max(abs(d) for d in diff) < 200  # sum of 3 digits x small errors / P + rounding: tiny next to q ~ 2^60
