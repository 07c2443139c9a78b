// CRT composition on the device, shared by the opcode-10 kernels (hevm_vm.hip, fused_ks.hip) and the decoder.
#pragma once
#include "modarith.hpp"

namespace dacapo {

struct CrtDev {
    const u64 *inv;  // [ell]       (q_0...q_{k-1})^{-1} mod q_k
    const u64 *mmod; // [ell][ell]  (q_0...q_{i-1}) mod q_k at [i*ell+k]
    const u64 *hmod; // [ell]       floor(Q/2) mod q_k
    const u64 *hdig; // [ell]       mixed-radix digits of floor(Q/2)
    const double *mdbl; // [ell]    (double)(q_0...q_{k-1})
};
constexpr int kMaxCrt = 32;

__device__ __forceinline__ double crt_centered(const u64 *__restrict__ coef, size_t n, int ell, size_t N,
                                      const DModulus *__restrict__ mods, const CrtDev c)
{
    u64 v[kMaxCrt];
    double acc = 0.0;
    for (int k = 0; k < ell; k++) {
        const DModulus M = mods[k];
        u64 s = 0;
        for (int i = 0; i < k; i++) s = addmod(s, mulmod(recanon(v[i], M), c.mmod[i * ell + k], M), M.q); // (v_i is a residue of q_i)
        const u64 y = addmod(coef[(size_t)k * N + n], c.hmod[k], M.q);
        v[k] = mulmod(submod(y, s, M.q), c.inv[k], M);
    }
    for (int k = ell - 1; k >= 0; k--) acc += (double)((long long)v[k] - (long long)c.hdig[k]) * c.mdbl[k];
    return acc;
}

// the same composition with the limb count known at compile time (no local array indexing left after unrolling)
template <int ELL>
__device__ __forceinline__ double crt_centered_fixed(const u64 *__restrict__ coef, size_t n, size_t N, const DModulus *__restrict__ mods,
                                                     const CrtDev c)
{
    u64 v[ELL];
#pragma unroll
    for (int k = 0; k < ELL; k++) {
        const DModulus M = mods[k];
        u64 s = 0;
#pragma unroll
        for (int i = 0; i < k; i++) s = addmod(s, mulmod(recanon(v[i], M), c.mmod[i * ELL + k], M), M.q);
        const u64 y = addmod(coef[(size_t)k * N + n], c.hmod[k], M.q);
        v[k] = mulmod(submod(y, s, M.q), c.inv[k], M);
    }
    double acc = 0.0;
#pragma unroll
    for (int k = ELL - 1; k >= 0; k--) acc += (double)((long long)v[k] - (long long)c.hdig[k]) * c.mdbl[k];
    return acc;
}

template <int ELL>
__device__ __forceinline__ double reencoded_coeff_fixed(const u64 *__restrict__ coef, size_t g, size_t N, const DModulus *__restrict__ mods,
                                                        const CrtDev c, double ratio)
{
    if (g == 0) return round(crt_centered_fixed<ELL>(coef, 0, N, mods, c) * ratio);
    if (g == N / 2) return 0.0;
    return round((crt_centered_fixed<ELL>(coef, g, N, mods, c) - crt_centered_fixed<ELL>(coef, N - g, N, mods, c)) * 0.5 * ratio);
}

// residue mod q of an integral double |x| < 2^120 (exact: at most 53 significant bits, shifted)
__device__ __forceinline__ u64 residue_of_double(double x, const DModulus &M)
{
    const bool neg = x < 0.0;
    const double a = fabs(x);
    u64 l, h;
    if (a < 0x1p63) {
        l = (u64)a;
        h = 0;
    } else {
        int e;
        const double fr = frexp(a, &e);
        const u64 mant = (u64)ldexp(fr, 53);
        const int sh = e - 53;
        l = sh < 64 ? mant << sh : 0;
        h = sh < 64 ? mant >> (64 - sh) : mant << (sh - 64);
    }
    const u64 r = reduce128_any(h, l, M);
    return (neg && r) ? M.q - r : r;
}

// coefficient g of the re-encoded plaintext of opcode 10 (see hevm_vm.hip): round((m_g - m_{N-g}) / 2 * ratio), m = centred
// CRT composition of coef[ell][N]; g = 0 keeps m_0, g = N/2 is 0.  (round() is odd, so coefficient N-g is the negative.)
__device__ __forceinline__ double reencoded_coeff(const u64 *__restrict__ coef, size_t g, int ell, size_t N, const DModulus *__restrict__ mods,
                                         const CrtDev c, double ratio)
{
    if (g == 0) return round(crt_centered(coef, 0, ell, N, mods, c) * ratio);
    if (g == N / 2) return 0.0;
    return round((crt_centered(coef, g, ell, N, mods, c) - crt_centered(coef, N - g, ell, N, mods, c)) * 0.5 * ratio);
}

} // namespace dacapo
