"""Real CKKS bootstrapping as an HEVM instruction sequence (scope row f4: what `bootstrap` means in the reference's HEaaN runtime,
/root/reference/lib/Runtime/HEAAN_HEVM.cpp:386-399 `bootstrapper->bootstrap(...)`, as opposed to the SEAL runtime's
decrypt / re-encrypt stand-in, SEAL_HEVM.cpp:324-334).  HEaaN is closed, so this is the published algorithm (Cheon-Han-Kim-Kim-Song
2018; Chen-Chillotti-Song 2019 / Han-Ki 2020 for the factored linear transforms and the double-angle sine), restated for SEAL's
conventions -- slot k = evaluation at zeta^(3^k), 60-bit primes, hybrid key switching -- and lowered onto the opcodes this runtime
already executes, plus four extension opcodes (hevm_asm.OP_ENCODE_COMPLEX / OP_CONJ / OP_MODRAISE / OP_SETSCALE):

    ModRaise      a ciphertext at 1 prime is read as one at L primes: it now decrypts to  t = p + q0 I,  |I| <~ sqrt(h)   (h = secret weight)
    CoeffToSlot   two ciphertexts whose SLOTS hold the coefficients t_j / q0 (bit-reversed order): u = A0^H z, u' = A0^H conj(D) z,
                  t_lo = 2 Re u / N, t_hi = 2 Re u' / N, where z = slots of t, A0[k][j] = zeta_k^j (j < N/2), D = diag(zeta_k^(N/2)) = +-i.
                  A0 = S_n ... S_4 S_2 P is a radix-2 FFT (S_m: butterflies at distance m/2 with twiddles zeta^((n/m) 3^k); with generator 3
                  the last factor is a 2x2 block [[1, w], [1, w^3]], w = e^(i pi/4), not a butterfly); each factor is three "diagonals"
                  (offsets 0, +-m/2) of a slot vector, a group of consecutive factors is one plaintext-matrix product (BSGS rotations).
    EvalMod       x = I + eps  ->  sin(2 pi x) ~ 2 pi eps:  cos(2 pi (x - 1/4) / 2^r) by its Taylor polynomial in theta^2, then r double
                  angles  c <- 2 c^2 - 1.  Every additive term carries an exactly tracked scale (constants are encoded with the
                  compensating factor), so the 2^-35 drift of each rescale (q = 2^60 - delta) never appears as an error.
    SlotToCoeff   z' = A0 y_lo + D A0 y_hi: the same factors in the other order, on the bit-reversed inputs EvalMod left.

Levels: 3 + 1 (CoeffToSlot) + 5 + r (EvalMod) + 3 (SlotToCoeff): `boot_levels`.  `simulate` interprets a program (all opcodes, the four extensions
included) on cleartext slot vectors with the exact scale semantics of the VM -- the reference semantics of the extension opcodes and
the CPU check of the whole construction (tests/test_ckks_boot.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import hevm_asm as ha
from .hevm_asm import (OP_ADDCC, OP_ADDCP, OP_BOOTSTRAP, OP_CONJ, OP_ENCODE, OP_ENCODE_COMPLEX, OP_MODRAISE, OP_MODSWITCH, OP_MULCC, OP_MULCP,
                       OP_NEGATE, OP_RESCALE, OP_ROTATE, OP_SETSCALE)


# ---- parameters ---------------------------------------------------------------------------------------------------------------
def _is_prime(n: int) -> bool:
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


def seal_prime_chain(logN: int, count: int, bits: int = 60):
    """CoeffModulus::Create(2^logN, {bits x count}) [SEAL-upstream]: scan down from 2^bits in steps of 2N; q_0 = last found, the
    special prime = first found (what dc_context_create builds; the scale bookkeeping below needs the exact values)"""
    step, found, v = 2 << logN, [], (1 << bits) + 1
    while len(found) < count:
        v -= step
        if _is_prime(v):
            found.append(v)
    return found[::-1]


def mixed_prime_chain(logN: int, widths):
    """one prime per entry of `widths` (bits), CoeffModulus::Create style per width class: scan down from 2^b in steps of 2N, no repeats, in
    the order given -- e.g. [60] + [51] * 30 + [60] * 7: a HEaaN-style chain, 60-bit base and special primes around 51-bit rescale primes
    (HEAAN_HEVM.cpp:55-56, profiled_HEAAN_GPU.json: rescalingFactor 51).  Every prime is 2^b - d, d < 2^28: what the runtime's arithmetic takes."""
    out, nxt = [], {}
    for b in widths:
        v = nxt.get(b, (1 << b) + 1)
        while True:
            v -= 2 << logN
            if _is_prime(v):
                break
        nxt[b] = v
        out.append(v)
    return out


def boot_levels(r: int = 5, groups: int = 3, taylor_terms: int = 16) -> int:
    """primes one bootstrap consumes: CoeffToSlot `groups` + 1, EvalMod 5 + r, SlotToCoeff `groups`"""
    return 2 * groups + 1 + 5 + r - (1 if taylor_terms == 8 else 0)


def _norm_off(o: int, n: int) -> int:
    o %= n
    return o - n if o > n // 2 else o


# ---- the special DFT as products of three-diagonal factors ----------------------------------------------------------------------
class DiagMatrix:
    """an n x n matrix given by its generalised diagonals:  (M x)[k] = sum_o d_o[k] x[(k + o) mod n]"""

    def __init__(self, n: int, diags: dict | None = None):
        self.n, self.d = n, {}
        for o, v in (diags or {}).items():
            self.add(o, v)

    def add(self, o: int, v):
        o = _norm_off(o, self.n)
        self.d[o] = self.d.get(o, 0) + np.asarray(v, dtype=np.complex128)

    def apply(self, x):
        return sum(v * np.roll(x, -o) for o, v in self.d.items())

    def after(self, first: "DiagMatrix") -> "DiagMatrix":
        """self * first (first is applied first)"""
        out = DiagMatrix(self.n)
        for a, da in self.d.items():
            for b, db in first.d.items():
                out.add(a + b, da * np.roll(db, -a))
        return out

    def times_diag_right(self, v) -> "DiagMatrix":  # self * diag(v)
        return DiagMatrix(self.n, {o: d * np.roll(v, -o) for o, d in self.d.items()})

    def times_diag_left(self, v) -> "DiagMatrix":  # diag(v) * self
        return DiagMatrix(self.n, {o: v * d for o, d in self.d.items()})

    def scaled(self, c) -> "DiagMatrix":
        return DiagMatrix(self.n, {o: c * d for o, d in self.d.items()})


def dft_factor(m: int, logN: int, herm: bool) -> DiagMatrix:
    """S_m (block size m) of A0 = S_n ... S_4 S_2 P, or its conjugate transpose"""
    N = 1 << logN
    n, M = N // 2, 2 * N
    k = np.arange(n) % m
    h = max(m // 2, 1)
    lower = k < h
    if m == 2:  # generator 3: 3^(n'/2) != N' + 1 at the last level, the block is [[1, w], [1, w^3]]
        w = np.exp(1j * np.pi / 4)
        if not herm:
            d0, dp, dm = np.where(lower, 1, w**3), np.where(lower, w, 0), np.where(lower, 0, 1)
        else:
            d0, dp, dm = np.where(lower, 1, np.conj(w**3)), np.where(lower, 1, 0), np.where(lower, 0, np.conj(w))
    else:
        e = np.array([pow(3, int(x), M) for x in range(h)], dtype=np.int64)
        tw = np.exp(2j * np.pi * (((n // m) * e) % M) / M)[np.where(lower, k, k - h)]
        if not herm:
            d0, dp, dm = np.where(lower, 1, -tw), np.where(lower, tw, 0), np.where(lower, 0, 1)
        else:
            d0, dp, dm = np.where(lower, 1, -np.conj(tw)), np.where(lower, 1, 0), np.where(lower, 0, np.conj(tw))
    out = DiagMatrix(n, {0: d0})
    out.add(h, dp)
    out.add(-h, dm)
    return out


def slot_exponents(logN: int):
    N = 1 << logN
    return np.array([pow(3, k, 2 * N) for k in range(N // 2)], dtype=np.int64)


def embed(coeffs, logN: int):
    """slots of a real polynomial: z_k = sum_j t_j zeta^(3^k j)  (one FFT of size 2N)"""
    N = 1 << logN
    full = np.fft.ifft(np.concatenate([np.asarray(coeffs, dtype=np.complex128), np.zeros(N)])) * (2 * N)  # sum_j t_j e^(+2 pi i r j / 2N)
    return full[slot_exponents(logN)]


def group_stages(logN: int, groups: int = 3):
    """block sizes m = 2 .. n split into `groups` runs of consecutive factors, smallest first"""
    ms = [1 << i for i in range(1, logN)]  # 2 .. n = N/2
    per = -(-len(ms) // groups)
    return [ms[i:i + per] for i in range(0, len(ms), per)]


# ---- emission onto a hevm_asm.Builder ---------------------------------------------------------------------------------------------
@dataclass
class Ct:
    v: ha.Value
    level: int
    s: float  # true scale: polynomial = s * (intended slot values), tracked exactly (the VM's own label may differ after addcp)


class BootstrapEmitter:
    """Emits the bootstrap of one ciphertext into `b`.  Plaintext registers of the matrices are shared by all bootstraps of a program."""

    def __init__(self, b: ha.Builder, logN: int, num_primes: int, target_level: int, r: int = 5, taylor_terms: int = 16, k_range: float = 16.0,
                 msg_bits: int = 0, diag_bits: int | None = None, out_bits: int = 40, groups: int = 3, cts_bits: int = 60, ks: int = 1, primes=None):
        """primes: the VM's chain when it is not CoeffModulus::Create(N, {60 x num_primes}) -- round 4: any chain of 45..60-bit primes, in
        particular a HEaaN-style mixed one (mixed_prime_chain).  Every scale below is tracked with the exact primes; the handful of places
        that used to say "60" now read the width of the prime they mean (self.qb)."""
        self.b, self.logN, self.N, self.n = b, logN, 1 << logN, 1 << (logN - 1)
        self.primes = [int(q) for q in primes] if primes is not None else seal_prime_chain(logN, num_primes)
        assert len(self.primes) == num_primes
        self.top = num_primes - ks  # ks special primes at the end of the chain (1 in SEAL's scheme; more with grouped-digit key switching)
        diag_bits = (self.qb(self.top) - 5) if diag_bits is None else diag_bits  # matrix / coefficient plaintexts: 55 bits next to 60-bit primes
        self.target, self.r, self.terms, self.k_range = target_level, r, taylor_terms, k_range
        self.msg_bits, self.diag_bits, self.out_bits, self.groups, self.cts_bits = msg_bits, diag_bits, out_bits, groups, cts_bits
        assert taylor_terms in (8, 16), "the polynomial in theta^2 is evaluated as a complete binary tree"
        # round 3: one more level than round 2 -- CoeffToSlot now ends with two rescales (see `bootstrap`)
        self.levels_needed = boot_levels(r, groups, taylor_terms)
        assert self.top - self.levels_needed == target_level, (
            f"{num_primes} primes leave {self.top - self.levels_needed} levels after a bootstrap, not {target_level}")
        self._plain_cache: dict = {}
        self._mats = None
        self.boot_in_bits = self.qb(1) - 10 - msg_bits  # scale of the ciphertext entering ModRaise: eps = p / q0 <= 2^-10 for |message| < 2^msg_bits

    def qb(self, level: int) -> int:
        """bits of the prime a rescale at `level` divides by"""
        return int(self.primes[level - 1]).bit_length()

    # -- low-level emitters (explicit levels, exact scales) -------------------------------------------------------------------
    def _val(self, level, s):
        return self.b._new(level, int(round(math.log2(s))) if s > 0 else 0, None)

    def _op(self, opcode, x: Ct, level, s, rhs=0, rhs_is_value=False) -> Ct:
        out = Ct(self._val(level, s), level, s)
        self.b._emit(opcode, out.v, x.v, rhs, rhs_is_value)
        return out

    def rotate(self, x: Ct, off: int) -> Ct:
        off = _norm_off(off, self.n)
        return x if off == 0 else self._op(OP_ROTATE, x, x.level, x.s, off & 0xFFFF)

    def conj(self, x: Ct) -> Ct:
        return self._op(OP_CONJ, x, x.level, x.s)

    def rescale(self, x: Ct) -> Ct:
        return self._op(OP_RESCALE, x, x.level - 1, x.s / float(self.primes[x.level - 1]))

    def modswitch(self, x: Ct, level: int) -> Ct:
        return x if level == x.level else self._op(OP_MODSWITCH, x, level, x.s, x.level - level)

    def add(self, x: Ct, y: Ct) -> Ct:
        assert x.level == y.level and abs(x.s / y.s - 1.0) < 1e-12, (x.level, y.level, x.s, y.s)
        return self._op(OP_ADDCC, x, x.level, y.s, y.v.id, True)

    def mul(self, x: Ct, y: Ct) -> Ct:
        assert x.level == y.level
        return self._op(OP_MULCC, x, x.level, x.s * y.s, y.v.id, True)

    def _plain_reg(self, key, vec, level, bits, complex_):
        k = (key, level, bits)
        if k not in self._plain_cache:
            b = self.b
            if complex_:
                vec = np.asarray(vec, dtype=np.complex128)
                idx = b._const(np.concatenate([vec.real, vec.imag]))
                reg = b.num_plain
                b.num_plain += 1
                b.ops.append(ha._Op(OP_ENCODE_COMPLEX, reg, idx, (level << 10) + bits, False, False))
            else:
                reg = b._encode(b._const(np.asarray(vec, dtype=np.float64)), level, bits)
            self._plain_cache[k] = reg
        return self._plain_cache[k]

    def mul_plain(self, x: Ct, key, vec, bits, complex_=True) -> Ct:
        reg = self._plain_reg(key, vec, x.level, bits, complex_)
        return self._op(OP_MULCP, x, x.level, x.s * 2.0**bits, reg)

    def add_const(self, x: Ct, c: float) -> Ct:
        """x + c exactly: the plaintext is encoded at a power of two near the true scale, with the value compensating the difference"""
        bits = int(round(math.log2(x.s)))
        val = c * x.s / 2.0**bits
        reg = self._plain_reg(("const", val), np.array([val]), x.level, bits, False)  # shared by every bootstrap of the program
        return self._op(OP_ADDCP, x, x.level, x.s, reg)

    def set_scale(self, x: Ct, label: float) -> Ct:
        idx = self.b._const(np.array([label, 0.0]))  # two entries: never confused with a scalar plaintext constant of the same value
        return self._op(OP_SETSCALE, x, x.level, x.s, idx)

    # -- matrices -----------------------------------------------------------------------------------------------------------------
    def matrices(self):
        if self._mats is None:
            logN, n = self.logN, self.n
            Dp = np.exp(2j * np.pi * ((slot_exponents(logN) * n) % (2 * self.N)) / (2 * self.N))  # zeta_k^n = +-i
            grp = group_stages(logN, self.groups)
            cts, stc = [], []
            for ms in reversed(grp):  # CoeffToSlot applies S_n^H first
                g = None
                for m in reversed(ms):
                    f = dft_factor(m, logN, herm=True)
                    g = f if g is None else f.after(g)
                cts.append(g)
            for ms in grp:  # SlotToCoeff applies S_2 first
                g = None
                for m in ms:
                    f = dft_factor(m, logN, herm=False)
                    g = f if g is None else f.after(g)
                stc.append(g)
            self._mats = {"cts": cts, "cts_hi_first": cts[0].times_diag_right(np.conj(Dp)), "stc": stc, "Dp": Dp}
        return self._mats

    def linear(self, x: Ct, mat: DiagMatrix, key: str, bits: int | None = None, baby_cache: dict | None = None, rescale: bool = True) -> Ct:
        """y = mat x by baby-step / giant-step, one plaintext product per diagonal, one rescale.  baby_cache: rotations of x by slot
        offset, shared between transforms of the same ciphertext (the two first-group CoeffToSlot matrices)"""
        bits = self.diag_bits if bits is None else bits
        offs = sorted(mat.d)
        g = 0
        for o in offs:
            g = math.gcd(g, abs(o))
        g = g or 1
        us = [o // g for o in offs]
        n1 = 1
        while n1 * n1 < len(us):
            n1 *= 2
        baby = {0: x}
        by_j: dict = {}
        for o, u in zip(offs, us):
            by_j.setdefault(u // n1, []).append((u % n1, o))
        total = None
        for j in sorted(by_j):
            G = g * n1 * j
            inner = None
            for i, o in sorted(by_j[j]):
                if i not in baby:
                    if baby_cache is not None and g * i in baby_cache:
                        baby[i] = baby_cache[g * i]
                    else:
                        baby[i] = self.rotate(x, g * i)
                        if baby_cache is not None:
                            baby_cache[g * i] = baby[i]
                term = self.mul_plain(baby[i], (key, o), np.roll(mat.d[o], G), bits)  # rot(d, -G)[k] = d[k - G]
                inner = term if inner is None else self.add(inner, term)
            inner = self.rotate(inner, G)
            total = inner if total is None else self.add(total, inner)
        return self.rescale(total) if rescale else total

    # -- EvalMod -------------------------------------------------------------------------------------------------------------------
    def eval_sine(self, x: Ct) -> Ct:
        """slots x = I + eps, |x| < k_range  ->  sin(2 pi x)"""
        r, T = self.r, self.terms
        xs = self.add_const(x, -0.25)
        kp = self.k_range + 0.25
        xs = Ct(xs.v, xs.level, xs.s * kp)                      # read the same polynomial as (x - 1/4) / kp in [-1, 1]: the powers of w stay
        w = self.rescale(self.mul(xs, xs))                     # below 1, so no Taylor coefficient drops under the plaintexts' 2^-55 grid
        pw = {1: w}
        k = 1
        while 2 * k < T:
            pw[2 * k] = self.rescale(self.mul(pw[k], pw[k]))
            k *= 2
        c2 = (2.0 * math.pi * kp / 2.0**r) ** 2               # theta^2 = c2 * w
        a = [(-1.0) ** i * c2**i / math.factorial(2 * i) for i in range(T)]  # cos(theta) = sum a_i w^i
        depth = int(math.log2(T))
        out_level = w.level - depth
        P = self._poly(a, pw, depth, out_level, 2.0 ** self.qb(out_level))  # a scale of one rescale prime: squaring + rescaling keeps it there
        for i in range(r):                                      # cos(2 t) = 2 cos^2 t - 1
            sq = self.mul(P, P)
            if i < r - 1:
                sq = self.rescale(sq)
                P = self.add_const(self.add(sq, sq), -1.0)
            else:
                # the last product stays at ~2^120: SlotToCoeff's baby-step rotations then act on a signal 2^60 above their
                # key-switch noise (sin(2 pi x) = 2 pi eps is tiny), and the rescale is paid after the last group instead.
                # The constant is added as -1/2 before the doubling: |-1 * 2^120| is the encoder's limit, 2^119 is not.
                h = self.add_const(sq, -0.5)
                P = self.add(h, h)
        return P                                                # cos(2 pi (x - 1/4)) = sin(2 pi x)

    def _poly(self, a, pw, k, out_level, S) -> Ct:
        """sum_{i < 2^k} a_i w^i at level `out_level` with true scale exactly S"""
        w = pw[1]
        if k == 1:
            q = float(self.primes[w.level - 1])
            kappa = S * q / (w.s * 2.0**self.diag_bits)
            t = self.rescale(self.mul_plain(w, ("coef", a[1] * kappa), np.array([a[1] * kappa]), self.diag_bits, complex_=False))
            t = Ct(t.v, t.level, S)  # = w.s 2^bits kappa / q by construction
            return self.add_const(self.modswitch(t, out_level), a[0])
        half = 1 << (k - 1)
        p = pw[half]
        q = float(self.primes[p.level - 1])
        hi = self._poly(a[half:], pw, k - 1, p.level, S * q / p.s)
        top = self.rescale(self.mul(hi, p))
        top = Ct(top.v, top.level, S)
        lo = self._poly(a[:half], pw, k - 1, out_level, S)
        return self.add(self.modswitch(top, out_level), lo)

    # -- the whole thing ---------------------------------------------------------------------------------------------------------
    def bootstrap(self, x: ha.Value, scale_in: float) -> tuple:
        """x: any level, label `scale_in` (its exact VM scale, ~2^40).  Returns (value at `target_level`, its exact label 2^out_bits)."""
        b, M = self.b, self.matrices()
        q0 = float(self.primes[0])
        ct = Ct(x, x.level, scale_in)
        ct = self.modswitch(ct, 1)
        up = self.boot_in_bits - int(round(math.log2(scale_in)))
        assert up >= 0, "the ciphertext entering a bootstrap must sit at or below 2^%d" % self.boot_in_bits
        if up:
            reg = b._encode(0xFFFF, 1, up)
            ct = self._op(OP_MULCP, ct, 1, ct.s * 2.0**up, reg)
        delta = ct.s                                             # p = delta * mu ; after ModRaise t = p + q0 I
        ct = self._op(OP_MODRAISE, ct, self.top, 1.0, self.top)  # from here s is relative to z = slots(t)
        # Noise budget (round 3; measured with the CPU oracle, tools/experiments/boot_precision.py).  What a bootstrap must preserve is
        # eps_j = p_j / q0 ~ 2^-(10 + msg_bits) / sqrt(N) per coefficient next to I_j ~ 2, so three absolute error sources that a
        # ciphertext at scale 2^40 would never notice decide the result: (a) key-switch noise of the baby-step rotations (~2^17 per slot
        # at N = 2^15, whatever the scale), (b) the integer rounding of the matrix plaintexts (relative 2^-(bits - 7) of |I|), (c) the
        # conjugation's key switch on the transform's output.  So: the raised ciphertext is multiplied by the integer 2^boost first
        # (exact: the all-ones "upscale" constant), the matrices are encoded at 2^60, and the transform ends with conj + add on the
        # un-rescaled sum followed by TWO rescales -- one level more than round 2, (a)-(c) pushed ~2^6 .. 2^40 further down.
        # (x - 1/4) / kp must enter EvalMod at a true scale of about one of ITS rescale primes (2^sw):
        #     2^(boost + sum cbits) N q0 kp / (the groups + 1 primes CoeffToSlot rescales by)  =  2^sw
        sw = self.qb(self.top - self.groups - 1)
        total = int(round(sum(self.qb(self.top - j) for j in range(self.groups + 1)) + sw - self.qb(1) - math.log2(self.k_range + 0.25) - self.logN))
        cbits = [min(self.qb(self.top - gi), self.cts_bits) for gi in range(self.groups)]
        boost = total - sum(cbits)
        assert 0 < boost < 60, boost
        reg = b._encode(0xFFFF, self.top, boost)
        ct = self._op(OP_MULCP, ct, self.top, ct.s * 2.0**boost, reg)
        # CoeffToSlot: lo = A0^H z, hi = A0^H conj(D) z.  boost and the plaintext scales are chosen so that (x - 1/4) / kp enters
        # EvalMod at a true scale of ~2^60 (every power of w then sits at ~2^60 too): 2^(boost + sum cbits) N q0 / (4 primes * kp) = 2^60
        shared: dict = {}                                       # both transforms rotate the same ciphertext by the same baby steps
        last = self.groups - 1
        lo = self.linear(ct, M["cts"][0], "cts0", cbits[0], shared, rescale=last != 0)
        hi = self.linear(ct, M["cts_hi_first"], "cts0h", cbits[0], shared, rescale=last != 0)
        for gi in range(1, self.groups):
            lo = self.linear(lo, M["cts"][gi], f"cts{gi}", cbits[gi], rescale=gi != last)
            hi = self.linear(hi, M["cts"][gi], f"cts{gi}", cbits[gi], rescale=gi != last)
        outs = []
        for u in (lo, hi):
            v = self.add(u, self.conj(u))                        # 2 Re u = N t  (slots now hold coefficients, bit-reversed)
            v = self.rescale(self.rescale(v))
            v = Ct(v.v, v.level, v.s * self.N * q0)              # ... read as x = t / q0
            outs.append(self.eval_sine(v))                       # sin(2 pi x) = 2 pi p / q0 (+ cubic error), at ~2^120
        # SlotToCoeff: z' = A0 y_lo + D A0 y_hi.  The last group carries kappa so that the result's label is exactly 2^out_bits.
        ylo, yhi = outs
        for gi in range(self.groups - 1):
            ylo = self.linear(ylo, M["stc"][gi], f"stc{gi}")
            yhi = self.linear(yhi, M["stc"][gi], f"stc{gi}")
        q_last, q_fin = float(self.primes[ylo.level - 1]), float(self.primes[ylo.level - 2])
        s_after = ylo.s * 2.0**self.diag_bits / (q_last * q_fin)  # true scale after the last group and the deferred rescale, without kappa
        # kappa is fixed for the NOMINAL input scale 2^boot_in_bits, so that every bootstrap of a program shares these plaintexts; the
        # instance's own scale (2^-35-level drift of the rescales before it) goes into the final label instead
        delta_nom = 2.0**self.boot_in_bits
        kappa = 2.0**self.out_bits * q0 / (s_after * 2.0 * math.pi * delta_nom)
        if "stcL" not in M:
            M["stcL"] = M["stc"][-1].scaled(kappa)
            M["stcLh"] = M["stcL"].times_diag_left(M["Dp"])
        zlo = self.linear(ylo, M["stcL"], "stcL", rescale=False)
        zhi = self.linear(yhi, M["stcLh"], "stcLh", rescale=False)
        z = self.rescale(self.rescale(self.add(zlo, zhi)))
        assert z.level == self.target, (z.level, self.target)
        label = 2.0**self.out_bits * (delta / delta_nom)
        out = self.set_scale(z, label)
        return out.v, label


# ---- exact scale labels of a Builder's values (what the VM computes at run time) -------------------------------------------------
class ScaleMirror:
    """value id -> the double the VM holds as that ciphertext's scale (SEAL_HEVM.cpp:268-334 semantics, in program order); incremental:
    `upto()` processes the instructions emitted since the last call"""

    def __init__(self, b: ha.Builder, primes):
        self.b, self.primes, self.done = b, primes, 0
        self.sc, self.lv, self.ps = {}, {}, {}

    def upto(self) -> dict:
        b, primes, sc, lv, ps = self.b, self.primes, self.sc, self.lv, self.ps
        for a in b.args:
            if a.id not in sc:
                sc[a.id], lv[a.id] = 2.0**a.scale_bits, a.level
        ops = b.ops
        for k in range(self.done, len(ops)):
            op = ops[k]
            if op.opcode in (OP_ENCODE, OP_ENCODE_COMPLEX):
                ps[op.dst] = 2.0 ** (op.rhs & 0x3FF)
                continue
            s, l = sc[op.lhs], lv[op.lhs]
            if op.opcode == OP_RESCALE:
                s, l = s / float(primes[l - 1]), l - 1
            elif op.opcode == OP_MODSWITCH:
                l -= op.rhs
            elif op.opcode == OP_ADDCC:
                s = sc[op.rhs]
            elif op.opcode == OP_ADDCP:
                s = ps[op.rhs]
            elif op.opcode == OP_MULCC:
                s *= sc[op.rhs]
            elif op.opcode == OP_MULCP:
                s *= ps[op.rhs]
            elif op.opcode == OP_BOOTSTRAP:
                s, l = 2.0 ** int(math.log2(s)), op.rhs
            elif op.opcode == OP_MODRAISE:
                l = op.rhs
            elif op.opcode == OP_SETSCALE:
                s = float(b.constants[op.rhs][0])
            sc[op.dst], lv[op.dst] = s, l
        self.done = len(ops)
        return sc


def vm_scales(b: ha.Builder, primes) -> dict:
    return ScaleMirror(b, primes).upto()


def _quantise(raw, logN: int):
    """what encoding does to a plaintext: its COEFFICIENTS are rounded to integers.  For a constant vector that is the value itself;
    for a general one the rounding noise is spread over the slots (modelled as the exact round trip only for small rings)"""
    if np.all(raw == raw[0]):
        return np.full_like(raw, np.round(raw[0].real) + 1j * np.round(raw[0].imag))
    return raw


# ---- cleartext interpreter with the VM's scale semantics -------------------------------------------------------------------------
def simulate(hevm: bytes, cst: bytes, inputs, logN: int, primes, secret_weight: int = 64, seed: int = 1, return_trace=False):
    """Runs a program on slot vectors: every register holds raw = slots(polynomial) (complex, length N/2) and the VM's scale label.
    ModRaise adds q0 * I for a random I distributed like <c1, s> / q0 for a ternary secret of the given weight; everything else is
    noise-free CKKS.  Returns the decoded results (raw / label)."""
    h = ha.unpack_hevm(hevm)
    consts = ha.unpack_cst(cst)
    n = 1 << (logN - 1)
    rng = np.random.default_rng(seed)
    idx = np.arange(n)
    reg, plain, trace = {}, {}, []
    for i, v in enumerate(inputs):
        v = np.asarray(v, dtype=np.float64).ravel()
        s = 2.0 ** h["arg_scale"][i]
        reg[i] = [(v[idx % len(v)] * s).astype(np.complex128), s, int(h["arg_level"][i])]
    for opc, dst, lhs, rhs in h["ops"].tolist():
        if opc == OP_ENCODE:
            s = 2.0 ** (rhs & 0x3FF)
            v = np.ones(1) if lhs == 0xFFFF else consts[lhs]
            plain[dst] = [_quantise((v[idx % len(v)] * s).astype(np.complex128), logN), s]
            continue
        if opc == OP_ENCODE_COMPLEX:
            s = 2.0 ** (rhs & 0x3FF)
            v = consts[lhs]
            half = len(v) // 2
            c = v[:half] + 1j * v[half:]
            plain[dst] = [_quantise(c[idx % half] * s, logN), s]
            continue
        if opc > OP_SETSCALE or opc == 5:
            continue
        raw, s, l = reg[lhs]
        if opc == OP_ROTATE:
            raw = np.roll(raw, -(rhs - 65536 if rhs >= 32768 else rhs))
        elif opc == OP_NEGATE:
            raw = -raw
        elif opc == OP_RESCALE:
            q = float(primes[l - 1])
            raw, s, l = raw / q, s / q, l - 1
        elif opc == OP_MODSWITCH:
            d = rhs - 65536 if rhs >= 32768 else rhs
            if d <= 0:
                continue
            l -= d
        elif opc == OP_ADDCC:
            raw, s = raw + reg[rhs][0], reg[rhs][1]
        elif opc == OP_ADDCP:
            raw, s = raw + plain[rhs][0], plain[rhs][1]
        elif opc == OP_MULCC:
            raw, s = raw * reg[rhs][0], s * reg[rhs][1]
        elif opc == OP_MULCP:
            raw, s = raw * plain[rhs][0], s * plain[rhs][1]
        elif opc == OP_BOOTSTRAP:
            raw, s, l = raw * (2.0 ** int(math.log2(s)) / s), 2.0 ** int(math.log2(s)), rhs
        elif opc == OP_CONJ:
            raw = np.conj(raw)
        elif opc == OP_MODRAISE:
            sigma = math.sqrt((secret_weight + 1) / 12.0)
            I = np.round(rng.normal(0.0, sigma, 2 * n))
            raw, l = raw + embed(I * float(primes[0]), logN), rhs
        elif opc == OP_SETSCALE:
            s = float(consts[rhs][0])
        reg[dst] = [raw, s, l]
        if return_trace:
            trace.append((opc, dst, l, s))
    outs = [reg[d][0] / reg[d][1] for d in h["res_dst"]]
    return (outs, trace) if return_trace else outs


# ---- a bootstrap on its own (tools/legs/boot_demo.py, bench.py, tests) ------------------------------------------------------------------
def single_bootstrap_program(logN: int, target: int = 3, r: int = 5, msg_bits: int = 0, ks: int = 1, primes=None):
    """(num_primes, cst, hevm, rotation offsets, emitter) of the program `one ciphertext at 1 prime, scale 2^40 -> bootstrap -> output`;
    ks = number of special primes of the chain (the VM must be created with ks_special = ks); primes: the VM's chain when it is not the
    all-60-bit one (target + boot_levels(r) + ks of them)"""
    K = target + boot_levels(r) + ks
    b = ha.Builder(slots=1 << (logN - 1), init_level=1, shadow=False)
    x = b.input(None, level=1, scale_bits=40)
    em = BootstrapEmitter(b, logN, K, target, r=r, msg_bits=msg_bits, ks=ks, primes=primes)
    y, _ = em.bootstrap(x, 2.0**40)
    b.output(y)
    cst, hv, _ = b.assemble()
    return K, cst, hv, rotation_offsets(hv), em


def rotation_offsets(hevm: bytes):
    """distinct non-zero slot offsets a program rotates by (for hevm_add_rotation_keys)"""
    return sorted({(int(q) - 65536 if q >= 32768 else int(q)) for o, _, _, q in ha.unpack_hevm(hevm)["ops"].tolist() if o == OP_ROTATE} - {0})


# ---- compiled programs: opcode 10 -> real bootstrapping ------------------------------------------------------------------------------
def lower_bootstraps(hevm: bytes, cst: bytes, logN: int, num_primes: int, msg_bits: int = 4, r: int = 5, ks: int = 1, primes=None):
    """Rewrites a program (e.g. one emitted by the reference's compiler) so that every opcode 10 -- `bootstrap`, a decrypt / re-encrypt
    stand-in in the SEAL runtime (SEAL_HEVM.cpp:324-334), the real thing in the HEaaN runtime (HEAAN_HEVM.cpp:386-399) -- becomes the
    real bootstrapping sequence of this module.  Everything else is re-emitted unchanged (same instructions, same constants, registers
    re-allocated).  All opcode 10 of the program must restore the same number of primes t, and the chain must hold num_primes =
    t + boot_levels(r) + ks primes (ks special ones).  Returns (hevm', cst')."""
    h = ha.unpack_hevm(hevm)
    consts = ha.unpack_cst(cst)
    slots = 1 << (logN - 1)
    qbits = [int(q).bit_length() for q in primes] if primes is not None else [60] * num_primes
    b = ha.Builder(slots=slots, init_level=int(h["init_level"]), shadow=False,
                   real_boot=dict(num_primes=num_primes, msg_bits=msg_bits, r=r, ks=ks, primes=primes))
    b.constants = [np.asarray(c, dtype=np.float64) for c in consts]
    b._const_index = {c.tobytes(): i for i, c in enumerate(b.constants)}
    cur = {}
    for i, (sb, lv) in enumerate(zip(h["arg_scale"], h["arg_level"])):
        cur[i] = b.input(None, level=int(lv), scale_bits=int(sb))
    plain_of, plain_bits = {}, {}
    for opc, dst, lhs, rhs in h["ops"].tolist():
        if opc in (OP_ENCODE, OP_ENCODE_COMPLEX):
            reg = b.num_plain
            b.num_plain += 1
            b.ops.append(ha._Op(opc, reg, lhs, rhs, False, False))
            plain_of[dst], plain_bits[dst] = reg, rhs & 0x3FF
            continue
        if opc > OP_SETSCALE or opc == 5 or (10 < opc < OP_CONJ):
            continue
        x = cur[lhs]
        lvl, bits = x.level, x.scale_bits
        if opc == OP_BOOTSTRAP:
            cur[dst] = b.bootstrap(x, int(rhs))
            continue
        if opc == OP_MODSWITCH:
            d = rhs - 65536 if rhs >= 32768 else rhs
            if d <= 0:
                continue
            lvl -= d
        elif opc == OP_RESCALE:
            lvl, bits = lvl - 1, bits - qbits[lvl - 1]
        elif opc == OP_ADDCC:
            bits = cur[rhs].scale_bits
        elif opc == OP_ADDCP:
            bits = plain_bits[rhs]
        elif opc == OP_MULCC:
            bits += cur[rhs].scale_bits
        elif opc == OP_MULCP:
            bits += plain_bits[rhs]
        elif opc == OP_MODRAISE:
            lvl = rhs
        out = b._new(lvl, bits, None)
        if opc in (OP_ADDCC, OP_MULCC):
            b._emit(opc, out, x, cur[rhs].id, True)
        elif opc in (OP_ADDCP, OP_MULCP):
            b._emit(opc, out, x, plain_of[rhs])
        else:
            b._emit(opc, out, x, rhs)
        cur[dst] = out
    for d in h["res_dst"]:
        b.output(cur[int(d)])
    cst2, hv2, _ = b.assemble()
    return hv2, cst2
