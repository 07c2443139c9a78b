// Device-side 64-bit modular arithmetic for the HEVM prime chain on gfx950 (CDNA4).
//
// CDNA4 has no 64-bit integer multiplier: every 64x64 product is built from v_mad_u64_u32
// (32x32+64 -> 64).  A Shoup/Harvey butterfly (what SEAL's CPU path uses) costs 10 of them; Barrett 11+.
// The HEVM chain is special, though: SEAL_HEVM.cpp:48-53 always asks CoeffModulus::Create for 60-bit
// primes, which SEAL picks by scanning DOWN from 2^60 in steps of 2N, so every prime is
//        q = 2^60 - delta,   delta < 2^28   (delta ~ 2^24.6 for the N=2^15 chain),
// and 2^60 == delta (mod q).  A 128-bit product therefore reduces with three small multiplies
// (64x28, 32x28) instead of a 64x64 high product + 64x64 low product: 7 v_mad_u64_u32 per modular
// multiply, no precomputed Shoup companion word (twiddle tables are half the size of SEAL's).
// All public results are canonical residues in [0,q), so limbs are bit-identical to any other correct
// implementation (SEAL's included) regardless of this lazy-reduction strategy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dacapo {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr int kQBits = 60;   // the reference's chain; the widest supported prime
constexpr int kMinQBits = 45; // round 3: any prime q = 2^b - d with 45 <= b <= 60, d < 2^28 and d 2^(64-b) < q (HEaaN-style 51-bit rescale primes included)
constexpr u32 kMaxDelta = 1u << 28;
// the second condition is what lets ONE fold take any 64-bit value below 2q (fold60 / canon): (x >> b) d + (x mod 2^b) < 2^(64-b) d + 2^b.
// It only bites at b = 45 (d < 2^26); tests/test_modarith_model.py executes the range arguments of this file on integers.
__host__ __device__ inline bool prime_shape_ok(u64 q, int bits)
{
    if (bits < kMinQBits || bits > kQBits) return false;
    const u64 d = (1ull << bits) - q;
    return d < kMaxDelta && (bits == kQBits || (d << (64 - bits)) < q); // (d < 2^28, 64 - b <= 19: no overflow)
}

// Per-prime constants, one entry per prime of the key-level chain, resident in HBM (and L2).
// `delta` is the WIDTH-TAGGED fold word every reduction below takes: bits 0..27 = d = 2^b - q, bits 28..31 = 60 - b.  For the
// reference's 60-bit chain the tag is 0 and the word is d itself.  The word is wave-uniform wherever a kernel handles one limb per
// workgroup, so the shift amount b - 32, the mask and d are scalar-unit values: the generic width costs the 60-bit path no vector
// instruction (the constant shifts of round 2 became register operands of the same v_alignbit / v_lshrrev).
struct DModulus {
    u64 q;
    u32 delta;   // width-tagged fold word (see above)
    u32 pad_;
    u64 inv_n;   // N^{-1} mod q
    u64 inv_n_w; // N^{-1} * (psi^{bitrev(1)})^{-1} mod q : last inverse-NTT stage twiddle with the scaling merged
};

__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * (u64)b + c; } // v_mad_u64_u32
__device__ __forceinline__ u32 lo32(u64 x) { return (u32)x; }
__device__ __forceinline__ u32 hi32(u64 x) { return (u32)(x >> 32); }
__device__ __forceinline__ u64 pack64(u32 lo, u32 hi) { return ((u64)hi << 32) | (u64)lo; }
// ({hi,lo} >> s) & 0xffffffff for 0 < s < 32 : one v_alignbit_b32
__device__ __forceinline__ u32 shr_pair(u32 hi, u32 lo, u32 s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
// Optimisation fence: hipcc otherwise re-associates the mad chains below (factoring out delta, splitting a 64-bit
// addend into a later v_lshl_add_u64) and spends 6-8 extra moves/adds per modular multiply.  Emits no instruction.
__device__ __forceinline__ u64 opaque(u64 x)
{
    asm("" : "+v"(x));
    return x;
}

__device__ __forceinline__ u32 opaque32(u32 x)
{
    asm("" : "+v"(x));
    return x;
}

// full 64x64 -> 128 product, 4 x v_mad_u64_u32 (any operands)
__device__ __forceinline__ void mul_wide(u64 a, u64 b, u64 &hi, u64 &lo)
{
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = mad32(a0, b1, p00 >> 32);
    u64 p10 = mad32(a1, b0, (u64)(u32)p01);
    u64 p11 = mad32(a1, b1, (p01 >> 32) + (p10 >> 32));
    lo = (p10 << 32) | (u64)(u32)p00;
    hi = p11;
}

// Two builds of the same sources (csrc/Makefile): libSEAL_HEVM.so with DC_GENERIC_WIDTH = 0 -- the reference's 60-bit chain only, every
// shift an immediate, exactly round 2's instruction streams -- and libSEAL_HEVM_gw.so with DC_GENERIC_WIDTH = 1, where the width travels
// in the fold word.  Measured with one build for both (run-time shifts and a uniform branch for the third fold inside every modular
// multiply): the 60-bit headline 46.2 -> 48.9 ms, the NTT leg 870 -> 1 030 us, config 3 458 -> 508 us; hence two builds.
#ifndef DC_GENERIC_WIDTH
#define DC_GENERIC_WIDTH 0
#endif
__host__ __device__ __forceinline__ u32 fold_word(int bits, u64 d) { return (u32)d | ((u32)(kQBits - bits) << 28); }
#if DC_GENERIC_WIDTH
__device__ __forceinline__ u32 fw_d(u32 fw) { return fw & 0x0FFFFFFFu; }          // 2^b - q
__device__ __forceinline__ u32 fw_sh(u32 fw) { return 28u - (fw >> 28); }         // b - 32
__device__ __forceinline__ u32 fw_mask(u32 fw) { return (1u << fw_sh(fw)) - 1u; } // low b - 32 bits of a high word
__device__ __forceinline__ bool fw_narrow(u32 fw) { return (fw >> 28) != 0; }
#else // the 60-bit chain: the tag is 0 (checked when the context is built), the word is d itself
__device__ __forceinline__ u32 fw_d(u32 fw) { return fw; }
__device__ __forceinline__ constexpr u32 fw_sh(u32) { return 28u; }
__device__ __forceinline__ constexpr u32 fw_mask(u32) { return 0x0FFFFFFFu; }
__device__ __forceinline__ constexpr bool fw_narrow(u32) { return false; }
#endif

// x (any 64-bit value) -> congruent value < 2^b + 2^(64-b) d < 2q.   1 mad   (named for the 60-bit chain it was written for)
__device__ __forceinline__ u64 fold60(u64 x, u32 fw)
{
    return mad32(hi32(x) >> fw_sh(fw), fw_d(fw), pack64(lo32(x), hi32(x) & fw_mask(fw)));
}

// any 64-bit value -> canonical residue
__device__ __forceinline__ u64 canon(u64 x, const DModulus &m)
{
    u64 r = fold60(x, m.delta);
    return r >= m.q ? r - m.q : r;
}

// T = hi*2^64 + w1*2^32 + w0 < 2^(2b+4) (products of lazy b-bit values; b = 60: < 2^124)  ->  congruent value < 4q.   3 mads, 5 other VALU ops
__device__ __forceinline__ u64 reduce_words(u64 hi, u32 w1, u32 w0, u32 fw)
{
    const u32 sh = fw_sh(fw), delta = fw_d(fw), mask = fw_mask(fw);
    const u32 H0 = shr_pair(lo32(hi), w1, sh);        // floor(T / 2^b), low word
    const u32 H1 = shr_pair(hi32(hi), lo32(hi), sh);   //                  high word (hi < 2^b)
    const u64 A = opaque(mad32(H0, delta, pack64(w0, w1 & mask))); // < 2^b + 2^b
    const u64 Bv = opaque((u64)H1 * delta);            // weight 2^32, < 2^b
    // Bv * 2^32 = (Bv >> (b-32)) * 2^b + (Bv & (2^(b-32) - 1)) * 2^32
    const u64 C = opaque(mad32(shr_pair(hi32(Bv), lo32(Bv), sh), delta, A)); // < 2^61 + 2^60
    // (fenced so that the sum stays a 32-bit add on the high word: hipcc otherwise turns it into v_and + v_mov 0 + a 64-bit add)
    const u64 R = pack64(lo32(C), opaque32(hi32(C) + (lo32(Bv) & mask))); // < 2^62: below 4q for the 60-bit chain (d ~ 2^25)
    // A narrower prime needs a third fold: each one divides by 2^b / d, and with b = 51, d ~ 2^26 two of them leave ~2^58.  The tag is
    // wave-uniform: the reference's chain skips this on the scalar unit.  Result then < 2^b + 2^(62-b) d < 2q.
    return fw_narrow(fw) ? fold60(R, fw) : R;
}
__device__ __forceinline__ u64 reduce128_lazy(u64 hi, u64 lo, u32 delta) { return reduce_words(hi, hi32(lo), lo32(lo), delta); }

// a < 2^60 (a twiddle, a key limb, a canonical residue), b < 2^63.  Result congruent to a*b, < 2^62.   7 mads.
// The middle term a0*b1 + a1*b0 (+ carry word) stays below 2^63 + 2^60 + 2^32, so it is one chained pair of mads
// with no 64-bit add and no carry handling.
__device__ __forceinline__ u64 mulmod_lazy(u64 a, u64 b, u32 delta)
{
    const u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    // (the two 32-bit carries are fenced BEFORE their zero extension: a fence on the extended 64-bit value makes hipcc copy the pair it
    // builds, v_mov_b32 + v_mov_b64 per carry -- 6 036 -> 5 686 VALU instructions per thread in the single-crossing NTT)
    const u64 p00 = opaque((u64)a0 * b0);
    const u64 mid = opaque(mad32(a1, b0, opaque(mad32(a0, b1, (u64)opaque32(hi32(p00))))));
    const u64 hi = opaque(mad32(a1, b1, (u64)opaque32(hi32(mid))));
    return reduce_words(hi, lo32(mid), lo32(p00), delta);
}

__device__ __forceinline__ u64 mulmod(u64 a, u64 b, const DModulus &m) { return canon(mulmod_lazy(a, b, m.delta), m); }

#if !DC_GENERIC_WIDTH
// Multiplication by a CONSTANT stored as the pair (w, W = w 2^31 mod q), both canonical: y < 2^62 -> congruent to w y, < 2q.   5 mads.
// With y = y0 + y1 2^31 (y0, y1 < 2^31): w y = w y0 + W y1 (mod q).  Column 0 = w.lo y0 + W.lo y1 < 2^63 + 2^63 and column 1 (weight 2^32) =
// w.hi y0 + W.hi y1 + carry < 2^59 + 2^59 (w.hi, W.hi < 2^28) cannot overflow, the sum T = col1 2^32 + lo32(col0) is below 2^92, so the
// part above 2^60 fits one word and ONE fold finishes: T mod 2^60 + (T >> 60) d <= 2^60 - 1 + (2^32 - 1) d < 2q for every d < 2^28.
__device__ __forceinline__ u64 mulmod_pair(u64 w, u64 W, u64 y, u32 delta)
{
    const u32 y0 = lo32(y) & 0x7FFFFFFFu, y1 = shr_pair(hi32(y), lo32(y), 31);
    const u64 c0 = opaque(mad32(lo32(W), y1, opaque((u64)lo32(w) * y0)));
    const u64 c1 = opaque(mad32(hi32(W), y1, opaque(mad32(hi32(w), y0, (u64)opaque32(hi32(c0))))));
    return opaque(mad32(shr_pair(hi32(c1), lo32(c1), 28), delta, pack64(lo32(c0), lo32(c1) & 0x0FFFFFFFu)));
}
#endif

// A canonical residue of ANOTHER prime of the chain (or the sum of two values below 2^60) -> canonical residue mod M.  Within one
// width class all primes lie within 2^28 of each other and one conditional subtraction is the whole reduction (SEAL: modulo_poly_coeffs
// only when q_j > q_m); a narrower target prime (mixed chains) needs the fold.  The width tag is wave-uniform.
__device__ __forceinline__ u64 recanon(u64 x, const DModulus &m)
{
    if (fw_narrow(m.delta)) return canon(x, m);
    return x >= m.q ? x - m.q : x;
}

// canonical operands
__device__ __forceinline__ u64 addmod(u64 a, u64 b, u64 q)
{
    u64 s = a + b;
    return s >= q ? s - q : s;
}
__device__ __forceinline__ u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
__device__ __forceinline__ u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

// any T = hi 2^64 + lo < 2^124 -> canonical residue (encoder, opcode 10: integers far beyond a product of two residues).  The 60-bit chain
// reduces directly; a narrower prime goes through (hi mod q) (2^64 mod q) + (lo mod q).
__device__ __forceinline__ u64 reduce128_any(u64 hi, u64 lo, const DModulus &m)
{
    if (fw_narrow(m.delta)) {
        const u64 r63 = canon(1ull << 63, m), r64 = r63 + r63 >= m.q ? r63 + r63 - m.q : r63 + r63;
        const u64 a = canon(mulmod_lazy(r64, canon(hi, m), m.delta), m), b = canon(lo, m);
        return a + b >= m.q ? a + b - m.q : a + b;
    }
    return canon(reduce128_lazy(hi, lo, m.delta), m);
}

// 128-bit accumulator for sums of products of canonical residues (each < 2^120): up to 16 terms keep
// hi < 2^60, the precondition of reduce128_lazy.
struct Acc128 {
    u64 hi, lo;
    __device__ __forceinline__ void clear() { hi = lo = 0; }
    __device__ __forceinline__ void mac(u64 a, u64 b)
    {
        u64 h, l;
        mul_wide(a, b, h, l);
        lo += l;
        hi += h + (lo < l);
    }
    __device__ __forceinline__ u64 reduce(const DModulus &m) const { return canon(reduce128_lazy(hi, lo, m.delta), m); }
};

} // namespace dacapo
