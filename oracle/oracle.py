"""ctypes binding + ciphertext-level composition + HEVM interpreter for the CPU oracle.

TEST INFRASTRUCTURE ONLY (see the header of ckks_oracle.c): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product package.

The C library restates SEAL 4.0.0's RNS-CKKS arithmetic at limb level; this file composes the limb
kernels into the evaluator calls that /root/reference/lib/Runtime/SEAL_HEVM.cpp makes
(one method per opcode handler, SEAL_HEVM.cpp:268-334) and interprets .hevm programs the way
SEAL_HEVM::run does (SEAL_HEVM.cpp:336-401).  Parity status: "parity unpinned" against SEAL itself.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import struct
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    """Compile oracle/libckks_oracle.so with gcc (recipe: oracle/Makefile)."""
    so = _HERE / "libckks_oracle.so"
    src = _HERE / "ckks_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "libckks_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        L = _LIB
        u64p = C.POINTER(C.c_uint64)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, u64p]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_psi.restype = C.c_uint64
        L.orc_psi.argtypes = [C.c_void_p, C.c_int]
        L.orc_min_primitive_root.restype = C.c_uint64
        L.orc_min_primitive_root.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_is_prime.argtypes = [C.c_uint64]
        L.orc_get_primes.argtypes = [C.c_uint64, C.c_int, C.c_int, u64p]
        L.orc_coeff_modulus_create.argtypes = [C.c_int, C.c_int, C.c_int, u64p]
        L.orc_elt_from_step.restype = C.c_uint32
        L.orc_elt_from_step.argtypes = [C.c_void_p, C.c_int]
        L.orc_bitrev.restype = C.c_uint32
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_encode.restype = C.c_int
        L.orc_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_void_p]
        L.orc_encode_complex.restype = C.c_int
        L.orc_encode_complex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_void_p]
        L.orc_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    return _LIB


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def splitmix_fill(seed: int, count: int) -> np.ndarray:
    """splitmix64 stream (seed 0x4845564D = "HEVM" by convention) as uint64[count]."""
    x = (np.arange(1, count + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed)
    with np.errstate(over="ignore"):
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


@dataclass
class Ciphertext:
    """[2][ell][N] uint64 limbs (NTT domain) + the double scale SEAL keeps per ciphertext."""
    data: np.ndarray
    scale: float

    @property
    def ell(self) -> int:
        return self.data.shape[1]

    def copy(self):
        return Ciphertext(self.data.copy(), self.scale)


@dataclass
class Plaintext:
    data: np.ndarray  # [ell][N]
    scale: float

    @property
    def ell(self) -> int:
        return self.data.shape[0]


@dataclass
class LazySum:
    """EXTENSION (the GPU VM's option hyb_lazy_sum, orc_rotate_acc_hybrid): a partial sum some of whose terms are grouped-digit rotations
    still in the raised basis.  base = the ordinary terms and the rotations' galois(c0); acc [2][ell + ks][N] = the rotations' inner products
    with their keys, not yet divided by P; have = rotations merged so far, of the group's `size`."""
    base: "Ciphertext"
    acc: np.ndarray
    group: int
    have: int
    size: int

    @property
    def ell(self) -> int:
        return self.base.ell

    @property
    def scale(self) -> float:
        return self.base.scale

    @scale.setter
    def scale(self, v):
        self.base.scale = v


class Oracle:
    def __init__(self, logN: int = 15, K: int = 14, bit_size: int = 60, primes=None):
        self.L = lib()
        arr = None
        if primes is not None:
            arr = (C.c_uint64 * len(primes))(*[int(p) for p in primes])
            K = len(primes)
        self.ctx = C.c_void_p(self.L.orc_create(logN, K, bit_size, arr))
        if not self.ctx:
            raise RuntimeError("orc_create failed")
        self.logN, self.N, self.K = logN, 1 << logN, K
        self.slots = self.N >> 1
        out = np.zeros(K, dtype=np.uint64)
        self.L.orc_primes(self.ctx, _p(out))
        self.primes = [int(x) for x in out]
        self.rng = C.c_uint64(0x4845564D)
        self.sk = self.pk = self.relin = None
        self.galois = {}
        self.ks, self.alpha = 1, 1  # SEAL: one special prime, one prime per digit

    def set_hybrid(self, ks: int, alpha: int | None = None):
        """EXTENSION (not SEAL): grouped-digit hybrid key switching -- the last `ks` primes are special, digits are groups of `alpha`
        data primes (default alpha = ks); keys become [dnum][2][K][N] (orc_keyswitch_hybrid).  set_hybrid(1, 1) is SEAL's scheme."""
        alpha = ks if alpha is None else alpha
        if self.L.orc_set_hybrid(self.ctx, ks, alpha) != 0:
            raise ValueError("bad hybrid parameters")
        self.ks, self.alpha = ks, alpha

    @property
    def max_level(self) -> int:
        return self.K - self.ks

    @property
    def dnum(self) -> int:
        return -(-(self.K - self.ks) // self.alpha)

    def __del__(self):
        try:
            self.L.orc_destroy(self.ctx)
        except Exception:
            pass

    # ---- tables / number theory --------------------------------------------------------------
    def root_powers(self, p: int) -> np.ndarray:
        out = np.zeros(self.N, dtype=np.uint64)
        self.L.orc_root_powers(self.ctx, p, _p(out))
        return out

    def inv_root_powers(self, p: int) -> np.ndarray:
        out = np.zeros(self.N, dtype=np.uint64)
        self.L.orc_inv_root_powers(self.ctx, p, _p(out))
        return out

    def psi(self, p: int) -> int:
        return int(self.L.orc_psi(self.ctx, p))

    # ---- NTT ---------------------------------------------------------------------------------
    def ntt_fwd(self, a: np.ndarray, pidx) -> np.ndarray:
        """a: [count][N]; limb b uses prime pidx[b]. Returns a new array."""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.N).copy()
        idx = np.ascontiguousarray(pidx, dtype=np.int32)
        assert len(idx) == a.shape[0]
        self.L.orc_ntt_fwd_batch(self.ctx, _p(idx), len(idx), _p(a))
        return a

    def ntt_inv(self, a: np.ndarray, pidx) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.N).copy()
        idx = np.ascontiguousarray(pidx, dtype=np.int32)
        self.L.orc_ntt_inv_batch(self.ctx, _p(idx), len(idx), _p(a))
        return a

    def ntt_fwd_simple(self, a, p):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.L.orc_ntt_fwd_simple(self.ctx, p, _p(a))
        return a

    def ntt_inv_simple(self, a, p):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.L.orc_ntt_inv_simple(self.ctx, p, _p(a))
        return a

    def ntt_fwd_definition(self, a, p):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.zeros_like(a)
        self.L.orc_ntt_fwd_definition(self.ctx, p, _p(a), _p(out))
        return out

    def negacyclic_schoolbook(self, a, b, p):
        out = np.zeros(self.N, dtype=np.uint64)
        self.L.orc_negacyclic_schoolbook(self.ctx, p, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
        return out

    # ---- limb-wise polynomial ops ---------------------------------------------------------------
    def _bin(self, fn, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.empty_like(a)
        fn(self.ctx, a.shape[0], _p(a), _p(b), _p(out))
        return out

    def poly_add(self, a, b):
        return self._bin(self.L.orc_poly_add, a, b)

    def poly_sub(self, a, b):
        return self._bin(self.L.orc_poly_sub, a, b)

    def poly_mul(self, a, b):
        return self._bin(self.L.orc_poly_mul, a, b)

    def poly_mul_simple(self, a, b):
        return self._bin(self.L.orc_poly_mul_simple, a, b)

    def poly_neg(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.empty_like(a)
        self.L.orc_poly_neg(self.ctx, a.shape[0], _p(a), _p(out))
        return out

    # ---- Galois ------------------------------------------------------------------------------
    def elt_from_step(self, step: int) -> int:
        e = int(self.L.orc_elt_from_step(self.ctx, int(step)))
        if e == 0:
            raise ValueError("step count too large")
        return e

    def default_galois_elts(self):
        out = np.zeros(2 * self.logN + 2, dtype=np.uint32)
        n = self.L.orc_default_galois_elts(self.ctx, _p(out))
        return [int(x) for x in out[:n]]

    def naf(self, v: int):
        out = np.zeros(40, dtype=np.int32)
        n = self.L.orc_naf(int(v), _p(out))
        return [int(x) for x in out[:n]]

    def galois_table(self, elt: int) -> np.ndarray:
        t = np.zeros(self.N, dtype=np.uint32)
        self.L.orc_galois_table(self.ctx, C.c_uint32(elt), _p(t))
        return t

    def galois_ntt(self, a: np.ndarray, elt: int) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.empty_like(a)
        self.L.orc_galois_ntt(self.ctx, C.c_uint32(elt), a.size // self.N, _p(a), _p(out))
        return out

    def galois_coeff(self, a: np.ndarray, elt: int, p: int) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.zeros_like(a)
        self.L.orc_galois_coeff(self.ctx, p, C.c_uint32(elt), _p(a), _p(out))
        return out

    # ---- rescale / key switch (limb level) -------------------------------------------------------
    def rescale_poly(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64)
        ell = a.shape[0]
        out = np.empty((ell - 1, self.N), dtype=np.uint64)
        self.L.orc_rescale_poly(self.ctx, ell, _p(a), _p(out))
        return out

    def divide_round_last(self, a: np.ndarray, pidx) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        idx = np.ascontiguousarray(pidx, dtype=np.int32)
        self.L.orc_divide_round_last(self.ctx, _p(idx), len(idx), _p(a))
        return a[:-1]

    def keyswitch(self, target: np.ndarray, key: np.ndarray, out0: np.ndarray, out1: np.ndarray):
        """Adds switch_key(target) into (out0, out1) in place. target/out: [ell][N]; key: [K-1][2][K][N]."""
        target = np.ascontiguousarray(target, dtype=np.uint64)
        if (self.ks, self.alpha) != (1, 1):
            assert key.flags["C_CONTIGUOUS"] and key.shape == (self.dnum, 2, self.K, self.N) and target.shape[0] <= self.max_level
            self.L.orc_keyswitch_hybrid(self.ctx, target.shape[0], _p(target), _p(key), _p(out0), _p(out1))
            return
        assert key.flags["C_CONTIGUOUS"] and key.shape == (self.K - 1, 2, self.K, self.N)
        self.L.orc_keyswitch(self.ctx, target.shape[0], _p(target), _p(key), _p(out0), _p(out1))

    def keyswitch_inner_simple(self, target, key):
        target = np.ascontiguousarray(target, dtype=np.uint64)
        ell = target.shape[0]
        out = np.zeros((2, ell + 1, self.N), dtype=np.uint64)
        self.L.orc_keyswitch_inner_simple(self.ctx, ell, _p(target), _p(key), _p(out))
        return out

    # ---- encoder ------------------------------------------------------------------------------
    def encode(self, values, scale: float, ell: int) -> Plaintext:
        """HEVM semantics (SEAL_HEVM.cpp:256-267): tile src[i % len] over N/2 slots, encode, keep ell primes."""
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        tiled = np.ascontiguousarray(v[np.arange(self.slots) % len(v)])
        out = np.zeros((ell, self.N), dtype=np.uint64)
        rc = self.L.orc_encode(self.ctx, _p(tiled), self.slots, float(scale), ell, _p(out))
        if rc:
            raise ValueError("encoded coefficient too large")
        return Plaintext(out, float(scale))

    def encode_complex(self, values, scale: float, ell: int) -> Plaintext:
        """complex slot values, tiled like encode (extension opcode 16: the DFT diagonals of real bootstrapping)"""
        v = np.ascontiguousarray(values, dtype=np.complex128).ravel()
        t = v[np.arange(self.slots) % len(v)]
        re, im = np.ascontiguousarray(t.real), np.ascontiguousarray(t.imag)
        out = np.zeros((ell, self.N), dtype=np.uint64)
        if self.L.orc_encode_complex(self.ctx, _p(re), _p(im), self.slots, float(scale), ell, _p(out)):
            raise ValueError("encoded coefficient too large")
        return Plaintext(out, float(scale))

    def modraise(self, a: Ciphertext, target: int) -> Ciphertext:
        """extension opcode 18: the centred residues modulo q_0 of a 1-prime ciphertext, read modulo the first `target` primes"""
        assert a.ell == 1
        q0 = np.uint64(self.primes[0])
        half = q0 >> np.uint64(1)
        out = np.zeros((2, target, self.N), dtype=np.uint64)
        for p in range(2):
            c = self.ntt_inv(a.data[p][:1], [0])[0]
            neg = c > half
            m = np.where(neg, q0 - c, c)
            for i in range(target):
                qi = np.uint64(self.primes[i])
                r = m % qi                                 # (a single conditional subtraction on the reference's chain; mixed chains: 60-bit residues into 51-bit primes)
                out[p, i] = np.where(neg & (r != 0), qi - r, r)
            out[p] = self.ntt_fwd(out[p], list(range(target)))
        return Ciphertext(out, a.scale)

    def keygen_sparse(self, weight: int, seed: int = 1, galois_elts=None):
        """like keygen, with a ternary secret of exactly `weight` non-zero coefficients (bootstrappable parameter sets)"""
        rng = np.random.default_rng(seed)
        coef = np.zeros(self.N, dtype=np.int64)
        pos = rng.choice(self.N, size=weight, replace=False)
        coef[pos] = rng.choice([-1, 1], size=weight)
        sk = np.stack([np.where(coef < 0, np.uint64(q - 1), coef.astype(np.uint64)) for q in self.primes])
        self.rng = C.c_uint64(seed)
        self.sk = self.ntt_fwd(sk, list(range(self.K)))
        self.pk = np.zeros((2, self.K, self.N), dtype=np.uint64)
        self.L.orc_gen_public(self.ctx, _p(self.sk), C.byref(self.rng), _p(self.pk))
        self.relin = self.gen_kswitch(self.poly_mul(self.sk, self.sk))
        self.galois = {}
        for elt in (self.default_galois_elts() if galois_elts is None else galois_elts):
            self.add_galois_key(elt)

    def decode(self, pt: Plaintext) -> np.ndarray:
        out = np.zeros(self.slots, dtype=np.float64)
        d = np.ascontiguousarray(pt.data)
        self.L.orc_decode(self.ctx, _p(d), d.shape[0], float(pt.scale), _p(out))
        return out

    # ---- keys / encrypt / decrypt ------------------------------------------------------------------
    def keygen(self, seed: int = 0x4845564D, galois_elts=None, relin: bool = True):
        """SEAL_HEVM::create_context's key set (SEAL_HEVM.cpp:60-83): sk, pk, relin, default Galois keys."""
        self.rng = C.c_uint64(seed)
        K, N = self.K, self.N
        self.sk = np.zeros((K, N), dtype=np.uint64)
        self.L.orc_gen_secret(self.ctx, C.byref(self.rng), _p(self.sk))
        self.pk = np.zeros((2, K, N), dtype=np.uint64)
        self.L.orc_gen_public(self.ctx, _p(self.sk), C.byref(self.rng), _p(self.pk))
        if relin:
            sk2 = self.poly_mul(self.sk, self.sk)
            self.relin = self.gen_kswitch(sk2)
        self.galois = {}
        for elt in (self.default_galois_elts() if galois_elts is None else galois_elts):
            self.add_galois_key(elt)

    def gen_kswitch(self, new_key: np.ndarray) -> np.ndarray:
        if (self.ks, self.alpha) != (1, 1):
            ksk = np.zeros((self.dnum, 2, self.K, self.N), dtype=np.uint64)
            self.L.orc_gen_kswitch_hybrid(self.ctx, _p(self.sk), _p(np.ascontiguousarray(new_key)), C.byref(self.rng), _p(ksk))
            return ksk
        ksk = np.zeros((self.K - 1, 2, self.K, self.N), dtype=np.uint64)
        self.L.orc_gen_kswitch(self.ctx, _p(self.sk), _p(np.ascontiguousarray(new_key)), C.byref(self.rng), _p(ksk))
        return ksk

    def add_galois_key(self, elt: int):
        self.galois[elt] = self.gen_kswitch(self.galois_ntt(self.sk, elt))

    def encrypt(self, pt: Plaintext) -> Ciphertext:
        ell = pt.ell
        out = np.zeros((2, ell, self.N), dtype=np.uint64)
        self.L.orc_encrypt(self.ctx, _p(self.pk), _p(np.ascontiguousarray(pt.data)), ell, C.byref(self.rng), _p(out))
        return Ciphertext(out, pt.scale)

    def decrypt(self, ct: Ciphertext) -> Plaintext:
        out = np.zeros((ct.ell, self.N), dtype=np.uint64)
        self.L.orc_decrypt(self.ctx, _p(self.sk), _p(np.ascontiguousarray(ct.data)), ct.ell, _p(out))
        return Plaintext(out, ct.scale)

    # ---- evaluator calls made by SEAL_HEVM.cpp (one per opcode) --------------------------------------
    def negate(self, a: Ciphertext) -> Ciphertext:  # SEAL_HEVM.cpp:275-279
        return Ciphertext(np.stack([self.poly_neg(a.data[0]), self.poly_neg(a.data[1])]), a.scale)

    def add(self, a: Ciphertext, b: Ciphertext) -> Ciphertext:  # SEAL_HEVM.cpp:297-303 (lhs.scale := rhs.scale)
        assert a.ell == b.ell
        return Ciphertext(np.stack([self.poly_add(a.data[0], b.data[0]), self.poly_add(a.data[1], b.data[1])]), b.scale)

    def add_plain(self, a: Ciphertext, p: Plaintext) -> Ciphertext:  # SEAL_HEVM.cpp:304-310
        assert a.ell == p.ell
        return Ciphertext(np.stack([self.poly_add(a.data[0], p.data), a.data[1].copy()]), p.scale)

    def mul_plain(self, a: Ciphertext, p: Plaintext) -> Ciphertext:  # SEAL_HEVM.cpp:318-323
        assert a.ell == p.ell
        return Ciphertext(np.stack([self.poly_mul(a.data[0], p.data), self.poly_mul(a.data[1], p.data)]), a.scale * p.scale)

    def tensor(self, a: Ciphertext, b: Ciphertext) -> np.ndarray:
        out = np.zeros((3, a.ell, self.N), dtype=np.uint64)
        self.L.orc_ct_tensor(self.ctx, a.ell, _p(np.ascontiguousarray(a.data)), _p(np.ascontiguousarray(b.data)), _p(out))
        return out

    def mul_relin(self, a: Ciphertext, b: Ciphertext) -> Ciphertext:  # SEAL_HEVM.cpp:311-317
        assert a.ell == b.ell
        t = self.tensor(a, b)
        c0, c1 = t[0].copy(), t[1].copy()
        self.keyswitch(t[2], self.relin, c0, c1)
        return Ciphertext(np.stack([c0, c1]), a.scale * b.scale)

    def apply_galois(self, a: Ciphertext, elt: int) -> Ciphertext:
        """Evaluator::apply_galois_inplace, CKKS branch.  Grouped-digit mode (extension): the digits are taken before the automorphism
        (orc_rotate_ks_hybrid), so that rotations of one ciphertext can share their decomposition."""
        if (self.ks, self.alpha) != (1, 1):
            c0 = self.galois_ntt(a.data[0], elt)
            c1 = np.zeros_like(c0)
            src1 = np.ascontiguousarray(a.data[1])
            key = self.galois[elt]
            assert key.shape == (self.dnum, 2, self.K, self.N)
            self.L.orc_rotate_ks_hybrid(self.ctx, a.ell, _p(src1), C.c_uint32(elt), _p(key), _p(c0), _p(c1))
            return Ciphertext(np.stack([c0, c1]), a.scale)
        c0 = self.galois_ntt(a.data[0], elt)
        temp = self.galois_ntt(a.data[1], elt)
        c1 = np.zeros_like(c0)
        self.keyswitch(temp, self.galois[elt], c0, c1)
        return Ciphertext(np.stack([c0, c1]), a.scale)

    def rotate_lazy(self, a: Ciphertext, steps: int, group: int, size: int) -> "LazySum":
        """one rotation as a term of a lazy sum: every hop but the last as usual, the last one left as (galois(c0), 0) and its inner products
        in the raised basis"""
        assert (self.ks, self.alpha) != (1, 1), "lazy sums exist in grouped-digit mode only"
        hops = self.rotate_hops(steps)
        assert hops, "a rotation by zero has no key switch to share"
        for e in hops[:-1]:
            a = self.apply_galois(a, e)
        elt = hops[-1]
        c0 = self.galois_ntt(a.data[0], elt)
        acc = np.zeros((2, a.ell + self.ks, self.N), dtype=np.uint64)
        key = self.galois[elt]
        self.L.orc_rotate_acc_hybrid(self.ctx, a.ell, _p(np.ascontiguousarray(a.data[1])), C.c_uint32(elt), _p(key), _p(acc))
        return LazySum(Ciphertext(np.stack([c0, np.zeros_like(c0)]), a.scale), acc, group, 1, size)

    def lazy_mul_plain(self, a: "LazySum", p: Plaintext, p_special: np.ndarray) -> "LazySum":
        """EXTENSION (the GPU VM's option hyb_double_hoist): a rotation still in the raised basis times a plaintext -- galois(c0) * p on the data
        primes as usual, the inner products * p limb by limb over the data primes AND the special primes (p_special [ks][N]: the same encoded
        polynomial reduced into them), so that the sum it joins is divided by P once"""
        assert a.ell == p.ell and p_special.shape == (self.ks, self.N) and a.have == 1
        ell = a.ell
        acc = np.empty_like(a.acc)
        sp = np.ascontiguousarray(p_special, dtype=np.uint64)
        for c in range(2):
            acc[c, :ell] = self.poly_mul(a.acc[c, :ell], p.data)
            hi, out = np.ascontiguousarray(a.acc[c, ell:]), np.empty((self.ks, self.N), dtype=np.uint64)
            self.L.orc_poly_mul_at(self.ctx, self.K - self.ks, self.ks, _p(hi), _p(sp), _p(out))
            acc[c, ell:] = out
        return LazySum(self.mul_plain(a.base, p), acc, a.group, a.have, a.size)

    def lazy_add(self, a, b):
        """a + b where either may be a LazySum of the same group; finished (one mod-down) when the group's last rotation has joined"""
        if not isinstance(a, LazySum):
            a, b = b, a
        if isinstance(b, LazySum):
            assert a.group == b.group, "two lazy sums met in one addition"
            ell = a.ell
            q = np.array(list(self.primes[:ell]) + list(self.primes[self.K - self.ks :]), dtype=np.uint64)[None, :, None]
            out = LazySum(self.add(a.base, b.base), (a.acc + b.acc) % q, a.group, a.have + b.have, a.size)
        else:
            out = LazySum(self.add(a.base, b), a.acc, a.group, a.have, a.size)
        if out.have < out.size:
            return out
        c0, c1 = out.base.data[0].copy(), out.base.data[1].copy()
        acc = np.ascontiguousarray(out.acc)
        self.L.orc_moddown_hybrid(self.ctx, out.ell, _p(acc), _p(c0), _p(c1))
        return Ciphertext(np.stack([c0, c1]), out.base.scale)

    rot_compose = False  # EXTENSION (the GPU VM's option rot_compose): see compose_rotation

    def compose_rotation(self, steps: int):
        """the GPU VM's HEVM::compose_rotation (hevm_vm.hip), restated: a rotation without a direct key as the SHORTEST sum (two parts, else
        three) of offsets that have one, candidates ascending by |offset|, positive before negative -- what lets a bounded key set serve every
        offset of a program, like the 49 left-rotation keys of the reference's HEaaN runtime (HEAAN_HEVM.cpp:58-64,124-126)"""
        slots = self.slots

        def norm(v):
            v %= slots
            if v > slots // 2:
                v -= slots
            if v <= -slots // 2:
                v += slots
            return v

        have, e, m = set(), 1, 2 * self.N
        for k in range(slots):
            if e in self.galois and norm(k) != 0:
                have.add(norm(k))
            e = (e * 3) % m
        order = sorted(have, key=lambda a: (abs(a), -a))
        t = norm(steps)
        for a in order:
            if norm(t - a) in have:
                return [a, norm(t - a)]
        for a in order:
            for b in order:
                if norm(t - a - b) in have:
                    return [a, b, norm(t - a - b)]
        return []

    def rotate_hops(self, steps: int):
        """Evaluator::rotate_internal's decomposition: list of galois elements applied in order."""
        if steps == 0:
            return []
        elt = self.elt_from_step(steps)
        if elt in self.galois:
            return [elt]
        if self.rot_compose:
            parts = self.compose_rotation(steps)
            if parts:
                return [self.elt_from_step(p) for p in parts]
        naf = self.naf(steps)
        if len(naf) == 1:
            raise KeyError("Galois key not present")
        hops = []
        for s in naf:
            if abs(s) != (self.N >> 1):
                hops += self.rotate_hops(s)
        return hops

    def rotate(self, a: Ciphertext, steps: int) -> Ciphertext:  # SEAL_HEVM.cpp:269-274
        out = a.copy()
        for elt in self.rotate_hops(steps):
            out = self.apply_galois(out, elt)
        return out

    def rescale(self, a: Ciphertext) -> Ciphertext:  # SEAL_HEVM.cpp:280-284
        q_last = self.primes[a.ell - 1]
        return Ciphertext(np.stack([self.rescale_poly(a.data[0]), self.rescale_poly(a.data[1])]), a.scale / float(q_last))

    def modswitch(self, a: Ciphertext, down: int) -> Ciphertext:  # SEAL_HEVM.cpp:285-293
        if down <= 0:
            return None  # dst untouched
        return Ciphertext(np.ascontiguousarray(a.data[:, : a.ell - down, :]), a.scale)

    def bootstrap(self, a: Ciphertext, target_level: int) -> Ciphertext:  # SEAL_HEVM.cpp:324-334
        vals = self.decode(self.decrypt(a))
        scale_bits = int(math.log2(a.scale))
        return self.encrypt(self.encode(vals, 2.0 ** scale_bits, target_level))


# ---- .hevm / .cst wire format (include/hecate/Support/HEVMHeader.h:10-35, SEAL_HEVM.cpp:182-234) ------
@dataclass
class Program:
    arg_scale: list
    arg_level: list
    res_scale: list
    res_level: list
    res_dst: list
    num_ctxt: int
    num_ptxt: int
    init_level: int
    ops: np.ndarray  # [n][4] uint16: opcode, dst, lhs, rhs


def read_hevm(path) -> Program:
    raw = Path(path).read_bytes()
    magic, hsize, arg_len, res_len = struct.unpack_from("<IIQQ", raw, 0)
    assert magic == 0x4845564D and hsize == 24
    body_len, nops, nct, npt, init_level = struct.unpack_from("<5Q", raw, 24)
    off = 64
    arrs = []
    for n in (arg_len, arg_len, res_len, res_len, res_len):
        arrs.append(list(struct.unpack_from(f"<{n}Q", raw, off)))
        off += 8 * n
    assert body_len == 40 + 8 * (2 * arg_len + 3 * res_len)
    ops = np.frombuffer(raw, dtype="<u2", count=4 * nops, offset=off).reshape(nops, 4).copy()
    return Program(*arrs, nct, npt, init_level, ops)


def read_cst(path):
    raw = Path(path).read_bytes()
    (n,) = struct.unpack_from("<q", raw, 0)
    off, out = 8, []
    for _ in range(n):
        (ln,) = struct.unpack_from("<q", raw, off)
        off += 8
        out.append(np.frombuffer(raw, dtype="<f8", count=ln, offset=off).copy())
        off += 8 * ln
    return out


class OracleVM:
    """SEAL_HEVM restated on the oracle: load / preprocess / encrypt / run / decrypt."""

    def __init__(self, oracle: Oracle):
        self.o = oracle
        self.prog = None
        self.consts = []
        self.ciphers = []
        self.plains = []

    def load(self, cst_path, hevm_path):
        self.consts = read_cst(cst_path)
        self.prog = read_hevm(hevm_path)
        self.ciphers = [None] * max(self.prog.num_ctxt, len(self.prog.arg_scale) + len(self.prog.res_dst))
        self.plains = [None] * self.prog.num_ptxt

    def encode_internal(self, src, level, scale_bits):  # SEAL_HEVM.cpp:256-267
        return self.o.encode(src, 2.0 ** scale_bits, level)

    def preprocess(self):  # SEAL_HEVM.cpp:242-254
        for opcode, dst, lhs, rhs in self.prog.ops:
            if opcode == 0:
                src = np.ones(1) if lhs == 0xFFFF else self.consts[lhs]
                self.plains[dst] = self.encode_internal(src, int(rhs) >> 10, int(rhs) & 0x3FF)
            elif opcode == 16:  # extension: complex constant stored as [re..., im...]
                v = self.consts[lhs]
                h = len(v) // 2
                self.plains[dst] = self.o.encode_complex(v[:h] + 1j * v[h:], 2.0 ** (int(rhs) & 0x3FF), int(rhs) >> 10)

    def encrypt(self, i, data):  # SEAL_HEVM.cpp:439-445
        pt = self.encode_internal(np.asarray(data, dtype=np.float64), self.prog.arg_level[i], self.prog.arg_scale[i])
        self.ciphers[i] = self.o.encrypt(pt)

    def decrypt(self, i):  # SEAL_HEVM.cpp:446-455
        return self.o.decode(self.o.decrypt(self.ciphers[i]))

    def decrypt_result(self, i):
        return self.decrypt(self.prog.res_dst[i])

    def set_lazy_groups(self, groups):
        """EXTENSION: the GPU plan's lazy sums (hevm_plan_lazy_groups): per group the instruction indices of the rotations whose mod-down
        is shared.  The fusion DECISION is the plan's; the arithmetic here is the oracle's own (Oracle.rotate_lazy / lazy_add)."""
        self.lazy = {}
        for g, ops in enumerate(groups):
            for i in ops:
                self.lazy[int(i)] = (g, len(ops))

    def step(self, op, index=None):
        o, c, p = self.o, self.ciphers, self.plains
        opcode, dst, lhs, rhs = (int(x) for x in op)
        lazy = getattr(self, "lazy", None)
        if lazy:
            if opcode == 1 and index in lazy:
                g, size = lazy[index]
                c[dst] = o.rotate_lazy(c[lhs], struct.unpack("<h", struct.pack("<H", rhs))[0], g, size)
                return
            if opcode == 6 and (isinstance(c[lhs], LazySum) or isinstance(c[rhs], LazySum)):
                c[lhs].scale = c[rhs].scale
                c[dst] = o.lazy_add(c[lhs], c[rhs])
                return
            if opcode == 9 and isinstance(c[lhs], LazySum) and c[lhs].have == 1 and rhs in getattr(self, "plains_special", {}):
                c[dst] = o.lazy_mul_plain(c[lhs], p[rhs], self.plains_special[rhs])  # (double hoisting: the plan multiplied inside the group)
                return
            for r in ((lhs,) if opcode in (1, 2, 3, 4, 7, 9, 10, 17, 18, 19) else (lhs, rhs) if opcode in (6, 8) else ()):
                if r < len(c) and isinstance(c[r], LazySum):
                    raise RuntimeError(f"op {index} (opcode {opcode}) reads an unfinished lazy sum: the plan's groups and the program disagree")
        if opcode == 1:
            c[dst] = o.rotate(c[lhs], struct.unpack("<h", struct.pack("<H", rhs))[0])
        elif opcode == 2:
            c[dst] = o.negate(c[lhs])
        elif opcode == 3:
            c[dst] = o.rescale(c[lhs])
        elif opcode == 4:
            r = o.modswitch(c[lhs], struct.unpack("<h", struct.pack("<H", rhs))[0])
            if r is not None:
                c[dst] = r
        elif opcode == 5:
            raise RuntimeError("This VM does not support native upscale op")
        elif opcode == 6:
            c[lhs].scale = c[rhs].scale
            c[dst] = o.add(c[lhs], c[rhs])
        elif opcode == 7:
            c[lhs].scale = p[rhs].scale
            c[dst] = o.add_plain(c[lhs], p[rhs])
        elif opcode == 8:
            c[dst] = o.mul_relin(c[lhs], c[rhs])
        elif opcode == 9:
            c[dst] = o.mul_plain(c[lhs], p[rhs])
        elif opcode == 10:
            c[dst] = o.bootstrap(c[lhs], rhs)
        elif opcode == 17:  # extension opcodes of this repo's VM (dacapo_amd/hevm_asm.py): conjugation, ModRaise, scale label
            c[dst] = o.apply_galois(c[lhs], 2 * o.N - 1)
        elif opcode == 18:
            c[dst] = o.modraise(c[lhs], rhs)
        elif opcode == 19:
            c[dst] = Ciphertext(c[lhs].data.copy(), float(self.consts[rhs][0]))
        # opcode 0 (encode) is a run-time no-op; 0xFFFF and unknown opcodes are no-ops (SEAL_HEVM.cpp:396-398)

    def run(self, max_ops=None):
        n = 0
        for op in self.prog.ops:
            if max_ops is not None and n >= max_ops:
                break
            self.step(op, n)
            n += 1
        return n
