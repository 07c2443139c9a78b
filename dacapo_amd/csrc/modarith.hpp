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

constexpr int kQBits = 60;
constexpr u64 kQMask = (1ull << kQBits) - 1;
constexpr u32 kMaxDelta = 1u << 28;

// Per-prime constants, one entry per prime of the key-level chain, resident in HBM (and L2).
struct DModulus {
    u64 q;
    u32 delta;   // 2^60 - q
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

// x (any 64-bit value) -> congruent value < 2^60 + 15*delta < 2q.   1 mad
__device__ __forceinline__ u64 fold60(u64 x, u32 delta)
{
    return mad32(hi32(x) >> 28, delta, pack64(lo32(x), hi32(x) & 0x0FFFFFFFu));
}

// any 64-bit value -> canonical residue
__device__ __forceinline__ u64 canon(u64 x, const DModulus &m)
{
    u64 r = fold60(x, m.delta);
    return r >= m.q ? r - m.q : r;
}

// T = hi*2^64 + w1*2^32 + w0 < 2^124  ->  congruent value < 2^62.   3 mads, 5 other VALU ops
__device__ __forceinline__ u64 reduce_words(u64 hi, u32 w1, u32 w0, u32 delta)
{
    const u32 H0 = shr_pair(lo32(hi), w1, 28);       // floor(T / 2^60), low word
    const u32 H1 = shr_pair(hi32(hi), lo32(hi), 28);  //                   high word (hi < 2^60)
    const u64 A = opaque(mad32(H0, delta, pack64(w0, w1 & 0x0FFFFFFFu))); // < 2^60 + 2^60
    const u64 Bv = opaque((u64)H1 * delta);           // weight 2^32, < 2^60
    // Bv * 2^32 = (Bv >> 28) * 2^60 + (Bv & (2^28-1)) * 2^32
    const u64 C = opaque(mad32(shr_pair(hi32(Bv), lo32(Bv), 28), delta, A)); // < 2^61 + 2^60
    return pack64(lo32(C), hi32(C) + (lo32(Bv) & 0x0FFFFFFFu));       // < 2^62
}
__device__ __forceinline__ u64 reduce128_lazy(u64 hi, u64 lo, u32 delta) { return reduce_words(hi, hi32(lo), lo32(lo), delta); }

// a < 2^60 (a twiddle, a key limb, a canonical residue), b < 2^63.  Result congruent to a*b, < 2^62.   7 mads.
// The middle term a0*b1 + a1*b0 (+ carry word) stays below 2^63 + 2^60 + 2^32, so it is one chained pair of mads
// with no 64-bit add and no carry handling.
__device__ __forceinline__ u64 mulmod_lazy(u64 a, u64 b, u32 delta)
{
    const u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    const u64 p00 = opaque((u64)a0 * b0);
    const u64 mid = opaque(mad32(a1, b0, opaque(mad32(a0, b1, opaque((u64)hi32(p00))))));
    const u64 hi = opaque(mad32(a1, b1, opaque((u64)hi32(mid))));
    return reduce_words(hi, lo32(mid), lo32(p00), delta);
}

__device__ __forceinline__ u64 mulmod(u64 a, u64 b, const DModulus &m) { return canon(mulmod_lazy(a, b, m.delta), m); }

// canonical operands
__device__ __forceinline__ u64 addmod(u64 a, u64 b, u64 q)
{
    u64 s = a + b;
    return s >= q ? s - q : s;
}
__device__ __forceinline__ u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
__device__ __forceinline__ u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

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
