// HEVM interpreter on the MI355X (see hevm_vm.hpp).  Restates struct SEAL_HEVM of
// /root/reference/lib/Runtime/SEAL_HEVM.cpp: loaders (:182-240), preprocess (:242-267), the opcode handlers
// (:268-334) and the dispatch loop (:336-401); key generation / encryption / decryption follow SEAL 4.0
// [SEAL-upstream keygenerator.cpp, rlwe.cpp, encryptor.cpp, decryptor.cpp, ckks.cpp].
#include "hevm_vm.hpp"

#include "../../include/dacapo_ckks.h"
#include "c_api_types.hpp"
#include "chacha.hpp"
#include "seal_serial.hpp"

#include <math.h>
#include <string.h>
#include <sys/random.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <fstream>
#include <iostream>

namespace dacapo {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
constexpr int kVmThreads = 256;

// ---------------------------------------------------------------------------------------------------------
// device kernels private to the VM: samplers, RNS lift, RLWE glue
// ---------------------------------------------------------------------------------------------------------
// Samplers: every thread owns 8 consecutive coefficients = one ChaCha20 block (chacha.hpp).
constexpr int kRngCoefs = 8;

__device__ __forceinline__ int rng_cbd(u64 w)
{ // sample_poly_cbd: 21 - 21 coin flips, sigma = 3.24 [SEAL-upstream clipnormal.h / rlwe.cpp]
    return __popcll(w & 0x1FFFFF) - __popcll((w >> 21) & 0x1FFFFF);
}
__device__ __forceinline__ void store_small8(u64 *__restrict__ out, const int (&v)[kRngCoefs], u64 q)
{
    u64x2 *o = reinterpret_cast<u64x2 *>(out);
#pragma unroll
    for (int e = 0; e < kRngCoefs; e += 2) {
        u64x2 t;
        t.x = v[e] < 0 ? q - (u64)(-v[e]) : (u64)v[e];
        t.y = v[e + 1] < 0 ? q - (u64)(-v[e + 1]) : (u64)v[e + 1];
        o[e >> 1] = t;
    }
}

// uniform residues mod q_i by rejection on 60-bit draws (sample_poly_uniform); limb i draws from object*64 + i.
// grid = (N/2048, limbs)
__global__ __launch_bounds__(kVmThreads) void sample_uniform_kernel(u64 *__restrict__ out, size_t N, ChaChaKey key, u64 object, u32 domain,
                                                                     const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y;
    const u64 q = mods[i].q;
    // draws of the prime's own width b (the fold word's tag is 60 - b: 0 on the reference's chain, where this is the 60-bit draw it always
    // was).  Round 3 drew 60 bits whatever the prime: for a 51-bit prime 255 retries almost never produced a residue, the key's uniform
    // halves came out non-canonical (and not uniform), and deep key switches on such keys left the lazy accumulators' range (round 4,
    // found by the mixed-chain bootstrap test).
    const u32 sh = 4 + (mods[i].delta >> 28);
    const size_t blk = (size_t)blockIdx.x * kVmThreads + threadIdx.x;
    u64 w[kRngCoefs], r[kRngCoefs];
    rng_words8(key, object * 64 + (u64)i, blk, 0, 0, domain, w);
#pragma unroll
    for (int e = 0; e < kRngCoefs; e++) r[e] = w[e] >> sh;
    for (u32 attempt = 1; attempt < 256; attempt++) { // a b-bit draw is >= q with probability d / 2^b (~2^-35 for the reference's primes): retry that word from a fresh block
        bool again = false;
#pragma unroll
        for (int e = 0; e < kRngCoefs; e++) again |= r[e] >= q;
        if (!again) break;
        rng_words8(key, object * 64 + (u64)i, blk, 0, attempt, domain, w);
#pragma unroll
        for (int e = 0; e < kRngCoefs; e++)
            if (r[e] >= q) r[e] = w[e] >> sh;
    }
    u64x2 *o = reinterpret_cast<u64x2 *>(out + (size_t)i * N + blk * kRngCoefs);
#pragma unroll
    for (int e = 0; e < kRngCoefs; e += 2) o[e >> 1] = u64x2{ r[e], r[e + 1] };
}

// one small signed polynomial (ternary: sample_poly_ternary; cbd: sample_poly_cbd) lifted to the first `limbs` primes.
// grid = (N/2048)
__global__ __launch_bounds__(kVmThreads) void sample_small_kernel(u64 *__restrict__ out, size_t N, int limbs, int cbd, ChaChaKey key,
                                                                   u64 object, u32 domain, const DModulus *__restrict__ mods)
{
    const size_t blk = (size_t)blockIdx.x * kVmThreads + threadIdx.x;
    u64 w[kRngCoefs];
    rng_words8(key, object, blk, 0, 0, domain, w);
    int v[kRngCoefs];
#pragma unroll
    for (int e = 0; e < kRngCoefs; e++) v[e] = cbd ? rng_cbd(w[e]) : rng_ternary(w[e]);
    for (int i = 0; i < limbs; i++) store_small8(out + (size_t)i * N + blk * kRngCoefs, v, mods[i].q);
}

// Encryptor::encrypt randomness of B encryptions in one launch: out[b][0] = ternary u, out[b][1], out[b][2] =
// centred-binomial e0, e1, each lifted to `cnt` primes.  Encryption b draws from object0 + b at the run() epoch kept in HBM
// (a replayed HIP graph still encrypts with fresh randomness).  grid = (N/2048, 3B)
__global__ __launch_bounds__(kVmThreads) void sample_enc_batch_kernel(u64 *__restrict__ out, size_t N, int cnt, ChaChaKey key, u64 object0,
                                                                       const DModulus *__restrict__ mods,
                                                                       const u64 *__restrict__ epoch)
{
    const int b = blockIdx.y / 3, z = blockIdx.y % 3;
    const size_t blk = (size_t)blockIdx.x * kVmThreads + threadIdx.x;
    u64 w[kRngCoefs];
    rng_words8(key, object0 + (u64)b, blk, *epoch, 0, RNG_ENC_U + (u32)z, w);
    int v[kRngCoefs];
#pragma unroll
    for (int e = 0; e < kRngCoefs; e++) v[e] = z ? rng_cbd(w[e]) : rng_ternary(w[e]);
    for (int i = 0; i < cnt; i++) store_small8(out + (((size_t)b * 3 + z) * cnt + i) * N + blk * kRngCoefs, v, mods[i].q);
}

// test hook: `blocks` consecutive ChaCha20 blocks starting at `counter` (16 words each)
__global__ void chacha_blocks_kernel(u32 *__restrict__ out, ChaChaKey key, u64 counter, u64 nonce, int blocks)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= blocks) return;
    u32 o[16];
    chacha20_block(key, counter + (u64)b, nonce, o);
    for (int i = 0; i < 16; i++) out[b * 16 + i] = o[i];
}

// signed 128-bit integer coefficients (two's complement, |x| < 2^120) -> residues.  grid = (N/256, ell)
__global__ void bump_epoch_kernel(u64 *epoch) { *epoch += 1; }

__global__ __launch_bounds__(kVmThreads) void lift_i128_kernel(u64 *__restrict__ out, const u64 *__restrict__ lo,
                                                                const u64 *__restrict__ hi, size_t N,
                                                                const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y;
    const DModulus M = mods[i];
    const size_t k = (size_t)blockIdx.x * kVmThreads + threadIdx.x;
    u64 l = lo[k], h = hi[k];
    const bool neg = (h >> 63) != 0;
    if (neg) {
        l = ~l + 1;
        h = ~h + (l == 0);
    }
    const u64 r = reduce128_any(h, l, M);
    out[(size_t)i * N + k] = neg ? negmod(r, M.q) : r;
}

// encrypt_zero_symmetric tail: c0 = -(c1*s + e) [+ (P mod q_i) * newkey on the limbs digit_lo <= i < digit_hi of the key's digit: one limb
// in SEAL's scheme, the digit's group with grouped digits].  grid = (N/512, limbs)
__global__ __launch_bounds__(kVmThreads) void ezs_final_kernel(u64 *__restrict__ c0, const u64 *__restrict__ c1,
                                                                const u64 *__restrict__ sk, const u64 *__restrict__ newkey,
                                                                int digit_lo, int digit_hi, const u64 *__restrict__ pmod, size_t N,
                                                                const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y;
    const DModulus M = mods[i];
    const size_t k = (size_t)i * N + ((size_t)blockIdx.x * kVmThreads + threadIdx.x) * 2;
    const u64x2 e = *reinterpret_cast<const u64x2 *>(c0 + k), a = *reinterpret_cast<const u64x2 *>(c1 + k),
                s = *reinterpret_cast<const u64x2 *>(sk + k);
    u64x2 r;
#pragma unroll
    for (int t = 0; t < 2; t++) r[t] = negmod(addmod(mulmod(a[t], s[t], M), e[t], M.q), M.q);
    if (newkey && i >= digit_lo && i < digit_hi) {
        const u64x2 nk = *reinterpret_cast<const u64x2 *>(newkey + k);
        const u64 factor = pmod[i];
#pragma unroll
        for (int t = 0; t < 2; t++) r[t] = addmod(r[t], mulmod(nk[t], factor, M), M.q);
    }
    *reinterpret_cast<u64x2 *>(c0 + k) = r;
}

// encrypt_zero_asymmetric body: tmp[p][i] = pk[p][i]*u[i] + e[p][i].  grid = (N/512, limbs, 2)
__global__ __launch_bounds__(kVmThreads) void pk_encrypt_kernel(u64 *__restrict__ tmp, long tmp_ps, const u64 *__restrict__ pk,
                                                                 long pk_ps, const u64 *__restrict__ u, const u64 *__restrict__ e01,
                                                                 long e_ps, size_t N, const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, p = blockIdx.z;
    const DModulus M = mods[i];
    const size_t k = (size_t)i * N + ((size_t)blockIdx.x * kVmThreads + threadIdx.x) * 2;
    const u64x2 e = *reinterpret_cast<const u64x2 *>(e01 + p * e_ps + k), a = *reinterpret_cast<const u64x2 *>(pk + p * pk_ps + k),
                uu = *reinterpret_cast<const u64x2 *>(u + k);
    u64x2 r;
#pragma unroll
    for (int t = 0; t < 2; t++) r[t] = addmod(mulmod(a[t], uu[t], M), e[t], M.q);
    *reinterpret_cast<u64x2 *>(tmp + p * tmp_ps + k) = r;
}

// Decryptor::decrypt (size 2): out = c0 + c1*s.  grid = (N/512, ell)
__global__ __launch_bounds__(kVmThreads) void decrypt_kernel(u64 *__restrict__ out, CtView ct, const u64 *__restrict__ sk,
                                                              size_t N, const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y;
    const DModulus M = mods[i];
    const size_t off = ((size_t)blockIdx.x * kVmThreads + threadIdx.x) * 2;
    const u64x2 c0 = *reinterpret_cast<const u64x2 *>(ct.limb(0, i, N) + off), c1 = *reinterpret_cast<const u64x2 *>(ct.limb(1, i, N) + off),
                s = *reinterpret_cast<const u64x2 *>(sk + (size_t)i * N + off);
    u64x2 r;
#pragma unroll
    for (int t = 0; t < 2; t++) r[t] = addmod(c0[t], mulmod(c1[t], s[t], M), M.q);
    *reinterpret_cast<u64x2 *>(out + (size_t)i * N + off) = r;
}


// ---- opcode 10 on the device -----------------------------------------------------------------------------------
// The SEAL VM's "bootstrap" (SEAL_HEVM.cpp:328-333) is decrypt -> decode -> encode(scale 2^floor(log2 scale),
// target level) -> encrypt.  decode keeps the REAL part of every slot and encode re-embeds it, so on the
// plaintext polynomial m (centred integer coefficients) the pair computes
//        m' = round( (m + m(X^-1)) / 2 * new_scale / old_scale ),     m(X^-1)_j = -m_{N-j}, m(X^-1)_0 = m_0,
// (the conjugate automorphism X -> X^{2N-1} is complex conjugation of the slots).  This kernel evaluates exactly that
// in one pass -- no FFT, no host round trip: CRT-compose each coefficient to a centred double with Garner's
// mixed-radix digits taken relative to the digits of floor(Q/2) (so small values cancel digit-wise instead of
// catastrophically), project, scale, round (half away from zero, like std::round in CKKSEncoder::encode) and emit
// the 128-bit two's-complement integer that lift_i128_kernel reduces into the target primes.
__device__ inline void store_i128(u64 *lo, u64 *hi, size_t idx, double x)
{ // x is integral, |x| < 2^120
    const bool neg = x < 0.0;
    const double a = fabs(x);
    u64 l, h;
    if (a < 0x1p63) {
        l = (u64)a;
        h = 0;
    } else {
        int e;
        const double fr = frexp(a, &e); // a = fr * 2^e, fr in [0.5, 1)
        const u64 mant = (u64)ldexp(fr, 53);
        const int sh = e - 53; // > 9
        l = sh < 64 ? mant << sh : 0;
        h = sh < 64 ? mant >> (64 - sh) : mant << (sh - 64);
    }
    if (neg) {
        l = ~l + 1;
        h = ~h + (l == 0);
    }
    lo[idx] = l;
    hi[idx] = h;
}

// grid = (N/2 + 1 threads).  coef: [ell][N] coefficient domain, canonical.
__global__ __launch_bounds__(kVmThreads) void reencode_kernel(u64 *__restrict__ lo, u64 *__restrict__ hi,
                                                               const u64 *__restrict__ coef, int ell, size_t N,
                                                               const DModulus *__restrict__ mods, const CrtDev c, double ratio)
{
    const size_t i = (size_t)blockIdx.x * kVmThreads + threadIdx.x;
    if (i > N / 2) return;
    if (i == 0) {
        store_i128(lo, hi, 0, round(crt_centered(coef, 0, ell, N, mods, c) * ratio));
    } else if (i == N / 2) {
        store_i128(lo, hi, i, 0.0);
    } else {
        const double a = crt_centered(coef, i, ell, N, mods, c), b = crt_centered(coef, N - i, ell, N, mods, c);
        const double r = round((a - b) * 0.5 * ratio);
        store_i128(lo, hi, i, r);
        store_i128(lo, hi, N - i, -r);
    }
}

// ---- batched opcode 10 (plan mode) -------------------------------------------------------------------------------
// Enc(pt) = Enc(0) + (pt, 0): the randomness-dependent half of every opcode 10 of the program does not depend on any
// ciphertext, so the plan makes all the zero-encryptions in a few large launches at the start of run() and the
// data-dependent half (decrypt -> re-encode -> NTT -> add) is 5 launches per batch of same-wave items.

// tmp[b][p][i] = pk[p][i]*u[b][i] + e[b][p][i]  (ue as written by sample_enc_batch_kernel, NTT form).  grid = (N/512, cnt, 2B)
__global__ __launch_bounds__(kVmThreads) void pk_encrypt_batch_kernel(u64 *__restrict__ tmp, const u64 *__restrict__ pk, long pk_ps,
                                                                       const u64 *__restrict__ ue, int cnt, size_t N,
                                                                       const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, b = blockIdx.z >> 1, p = blockIdx.z & 1;
    const DModulus M = mods[i];
    const size_t k = ((size_t)blockIdx.x * kVmThreads + threadIdx.x) * 2;
    const u64 *u = ue + (((size_t)b * 3) * cnt + i) * N + k, *e = ue + (((size_t)b * 3 + 1 + p) * cnt + i) * N + k;
    const u64x2 ev = *reinterpret_cast<const u64x2 *>(e), a = *reinterpret_cast<const u64x2 *>(pk + p * pk_ps + (size_t)i * N + k),
                uu = *reinterpret_cast<const u64x2 *>(u);
    u64x2 r;
#pragma unroll
    for (int t = 0; t < 2; t++) r[t] = addmod(mulmod(a[t], uu[t], M), ev[t], M.q);
    *reinterpret_cast<u64x2 *>(tmp + (((size_t)b * 2 + p) * cnt + i) * N + k) = r;
}

__device__ inline void store_residues(u64 *__restrict__ out, size_t idx, size_t N, int t, double x, bool flip,
                                      const DModulus *__restrict__ mods)
{ // x integral, |x| < 2^120: writes (flip ? -x : x) mod q_k for k < t
    const bool neg = (x < 0.0) != flip;
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
    for (int k = 0; k < t; k++) {
        const DModulus M = mods[k];
        const u64 r = canon(reduce128_lazy(h, l, M.delta), M);
        out[(size_t)k * N + idx] = (neg && r) ? M.q - r : r;
    }
}

// reencode_kernel + lift_i128_kernel for a batch: coef [B][ell][N] (coefficient domain, canonical) -> ptx [B][t][N]
// (coefficient domain residues of the re-encoded plaintext).  grid = (ceil((N/2+1)/256), B)
__global__ __launch_bounds__(kVmThreads) void reencode_lift_batch_kernel(u64 *__restrict__ ptx, const u64 *__restrict__ coef,
                                                                          const BootItem *__restrict__ items, int ell, int t, size_t N,
                                                                          const DModulus *__restrict__ mods, const CrtDev c)
{
    const size_t i = (size_t)blockIdx.x * kVmThreads + threadIdx.x;
    if (i > N / 2) return;
    const int b = blockIdx.y;
    const double ratio = items[b].ratio;
    const u64 *cf = coef + (size_t)b * ell * N;
    u64 *out = ptx + (size_t)b * t * N;
    if (i == 0) {
        store_residues(out, 0, N, t, round(crt_centered(cf, 0, ell, N, mods, c) * ratio), false, mods);
    } else if (i == N / 2) {
        store_residues(out, i, N, t, 0.0, false, mods);
    } else {
        const double a = crt_centered(cf, i, ell, N, mods, c), bb = crt_centered(cf, N - i, ell, N, mods, c);
        const double r = round((a - bb) * 0.5 * ratio);
        store_residues(out, i, N, t, r, false, mods);
        store_residues(out, N - i, N, t, r, true, mods);
    }
}

// decode, first half (CKKSEncoder::decode_internal): compose every coefficient from its residues, centre it, divide by the scale.
// grid = (N/256)
__global__ __launch_bounds__(kVmThreads) void dec_crt_kernel(double2 *__restrict__ v, const u64 *__restrict__ coef, int ell, size_t N,
                                                              const DModulus *__restrict__ mods, const CrtDev c, double inv_scale)
{
    const size_t n = (size_t)blockIdx.x * kVmThreads + threadIdx.x;
    v[n] = make_double2(crt_centered(coef, n, ell, N, mods, c) * inv_scale, 0.0);
}

// ---------------------------------------------------------------------------------------------------------
// CKKSEncoder on the host  [SEAL-upstream ckks.cpp, dwthandler.h]
// ---------------------------------------------------------------------------------------------------------
HostEncoder::HostEncoder(int logN_) : N((size_t)1 << logN_), slots(N >> 1), logN(logN_)
{
    root_.resize(N);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (size_t k = 0; k < N; k++) {
        const long double ang = two_pi * (long double)h_bitrev((u32)k, logN) / (long double)(2 * N);
        root_[k] = std::complex<double>((double)cosl(ang), (double)sinl(ang));
    }
    slot_map_.resize(N);
    const u64 m = 2 * (u64)N;
    u64 pos = 1;
    for (size_t i = 0; i < slots; i++) {
        slot_map_[i] = h_bitrev((u32)((pos - 1) >> 1), logN);
        slot_map_[slots | i] = h_bitrev((u32)((m - pos - 1) >> 1), logN);
        pos = (pos * 3) & (m - 1);
    }
}

void HostEncoder::encode(const double *src, size_t len, double scale, std::vector<__int128> &coeffs) const
{
    std::vector<std::complex<double>> v(N, 0.0);
    for (size_t i = 0; i < slots; i++) {
        const double x = src[i % len];
        v[slot_map_[i]] = x;
        v[slot_map_[slots | i]] = x; // conjugate of a real value
    }
    // transform_from_rev with inverse roots; the scalar (scale / n) rides on the last stage
    for (size_t m = N >> 1, gap = 1; m > 1; m >>= 1, gap <<= 1)
        for (size_t i = 0; i < m; i++) {
            const std::complex<double> r = std::conj(root_[m + i]);
            std::complex<double> *x = v.data() + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                const std::complex<double> a = x[j], b = y[j];
                x[j] = a + b;
                y[j] = (a - b) * r;
            }
        }
    const double fix = scale / (double)N;
    {
        const size_t gap = N >> 1;
        const std::complex<double> sr = std::conj(root_[1]) * fix;
        for (size_t j = 0; j < gap; j++) {
            const std::complex<double> a = v[j], b = v[j + gap];
            v[j] = (a + b) * fix;
            v[j + gap] = (a - b) * sr;
        }
    }
    coeffs.resize(N);
    for (size_t j = 0; j < N; j++) {
        const double c = round(v[j].real());
        if (!(fabs(c) < 0x1p120)) {
            fprintf(stderr, "[dacapo_amd] encode: coefficient does not fit 120 bits (scale too large)\n");
            abort();
        }
        coeffs[j] = (__int128)c; // exact: |c| < 2^120 is representable after the double -> int128 conversion
    }
}

void HostEncoder::decode(std::vector<std::complex<double>> &v, double *out) const
{
    for (size_t m = 1, gap = N >> 1; m < N; m <<= 1, gap >>= 1)
        for (size_t i = 0; i < m; i++) {
            const std::complex<double> r = root_[m + i];
            std::complex<double> *x = v.data() + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                const std::complex<double> a = x[j], b = y[j] * r;
                x[j] = a + b;
                y[j] = a - b;
            }
        }
    for (size_t i = 0; i < slots; i++) out[i] = v[slot_map_[i]].real();
}

// ---------------------------------------------------------------------------------------------------------
// context + keys
// ---------------------------------------------------------------------------------------------------------
RngKeys rng_keys_from_os()
{
    RngKeys k;
    unsigned char buf[sizeof(RngKeys)];
    size_t got = 0;
    while (got < sizeof(buf)) {
        const ssize_t r = getrandom(buf + got, sizeof(buf) - got, 0);
        if (r <= 0) {
            fprintf(stderr, "[dacapo_amd] getrandom() failed: no cryptographic randomness available, refusing to generate keys or encrypt\n");
            abort();
        }
        got += (size_t)r;
    }
    memcpy(&k, buf, sizeof(k));
    return k;
}


static u64 *dalloc(size_t elems)
{
    u64 *p = nullptr;
    DC_HIP_CHECK(vm_malloc(&p, elems * sizeof(u64)));
    return p;
}

void HEVM::init_context(int logN, int K, const u64 *primes, int dir_ksp, int dir_alpha)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "[dacapo_amd] no HIP device: the HEVM runtime has no CPU fallback\n");
        abort();
    }
    // VM options (options.hpp), read once here: this VM keeps what it was created with.
    // EXTENSION: ks_special = k > 1 switches key switching to grouped digits (hybrid_ks.hip): the last k primes of the chain are special,
    // a digit is ks_alpha (default k) data primes.  Not SEAL's key format: a key directory written in this mode records (k, alpha) in
    // hybrid.txt and loads only into a VM created with the same two values.
    int ksp = std::max(1, (int)option(OPT_KS_SPECIAL));
    int alpha = option(OPT_KS_ALPHA) > 0 ? (int)option(OPT_KS_ALPHA) : ksp;
    if (dir_ksp > 0) { // a key directory written in grouped-digit mode says so itself (hybrid.txt): it sets the mode unless the options ask for another
        if ((ksp != 1 || alpha != 1) && (ksp != dir_ksp || alpha != dir_alpha)) {
            fprintf(stderr, "[dacapo_amd] the key directory was written with ks_special = %d, ks_alpha = %d; options ask for %d / %d\n", dir_ksp,
                    dir_alpha, ksp, alpha);
            abort();
        }
        ksp = dir_ksp, alpha = dir_alpha;
    }
    // prime_bits = b (45..60; the generic-width build only): the chain CoeffModulus::Create(N, {b, b, ...}) instead of the reference's
    // 60-bit one (SEAL_HEVM.cpp:48-53) -- e.g. 51 for rescale primes of the HEaaN configuration's width
    const int bits = (int)option(OPT_PRIME_BITS);
    ctx.reset(new Context(logN, K, bits, primes, ksp, alpha));
    ctx->ensure_scratch();
    encoder.reset(new HostEncoder(logN));
    use_plan = option(OPT_PLAN) != 0;
    plan_graph = option(OPT_PLAN_GRAPH) != 0;
    plan_dag = option(OPT_PLAN_GRAPH) == 2;
    plan_lanes = option(OPT_PLAN_LANES) >= 2 ? 2 : 1;
    host_encoder = option(OPT_HOST_ENCODER) != 0;
    fold_rescale_into_boot = option(OPT_FOLD_RESCALE_BOOT) != 0;
    lazy_sums = option(OPT_HYB_LAZY_SUM) != 0;
    double_hoist = lazy_sums && option(OPT_HYB_DOUBLE_HOIST) != 0;
    max_batch = std::max(1, (int)option(OPT_MAX_BATCH));
    chain_fusion = option(OPT_CHAIN_FUSION) != 0;
    secret_weight = (int)option(OPT_SECRET_HW);
    rot_compose = option(OPT_ROT_COMPOSE) != 0;
    online_encode = option(OPT_ONLINE_ENCODE) != 0 && use_plan && !host_encoder;
    lanes.resize(1);
    DC_HIP_CHECK(hipStreamCreateWithFlags(&lanes[0].stream, hipStreamNonBlocking));
    lanes[0].ws = ctx->ws0;
    lanes[0].boot_plain.d = dalloc((size_t)ctx->max_level() * ctx->N);
    lanes[0].boot_plain.level = ctx->max_level();
    DC_HIP_CHECK(vm_malloc(&d_epoch, 8));
    DC_HIP_CHECK(hipMemset(d_epoch, 0, 8));
    cur = 0;
}

// KeyGenerator::generate_one_kswitch_key for every digit: key[j] = (-(a_j s + e_j) + [limb j](P mod q_j) s', a_j)
void HEVM::gen_kswitch_key(u64 *key, const u64 *new_key, u64 key_id)
{
    Context &c = *ctx;
    const size_t N = c.N;
    const int K = c.K;
    const dim3 gu((unsigned)(N / (kRngCoefs * kVmThreads)), (unsigned)K), gs((unsigned)(N / (kRngCoefs * kVmThreads))),
        g2((unsigned)(N / (2 * kVmThreads)), (unsigned)K);
    const int L = c.max_level();
    for (int j = 0; j < c.key_digits(); j++) {
        u64 *c0 = key + (size_t)j * 2 * K * N, *c1 = c0 + (size_t)K * N;
        const u64 object = key_id * 64 + (u64)j; // one object per (key, digit)
        DC_LAUNCH(sample_uniform_kernel, gu, dim3(kVmThreads), 0, S(), c1, N, rng.pub, object, (u32)RNG_KSK_A, c.d_mods);
        DC_LAUNCH(sample_small_kernel, gs, dim3(kVmThreads), 0, S(), c0, N, K, 1, rng.secret, object, (u32)RNG_KSK_E, c.d_mods);
        launch_ntt(c, false, c0, (long)N, K, nullptr, 0, 0, S());
        const int lo = c.hybrid() ? j * c.alpha : j, hi = c.hybrid() ? std::min(lo + c.alpha, L) : j + 1;
        DC_LAUNCH(ezs_final_kernel, g2, dim3(kVmThreads), 0, S(), c0, c1, keys.sk, new_key, lo, hi, c.d_pmod, N, c.d_mods);
    }
}

void HEVM::add_galois_key(u32 elt)
{
    if (keys.galois.count(elt)) return;
    Context &c = *ctx;
    u64 *rot = W().ct_tmp; // [K][N] fits: scratch is 3*(K-1)*N
    launch_galois(c, CtView{ rot, 0 }, CtView{ keys.sk, 0 }, elt, 1, c.K, S());
    u64 *key = dalloc(key_elems());
    gen_kswitch_key(key, rot, 16 + (u64)elt); // key ids: 8 = relinearisation, 16 + elt = Galois element
    keys.galois[elt] = key;
}

// SEAL_HEVM::create_context's key set (SEAL_HEVM.cpp:60-83): secret, public, relin, default Galois keys
void HEVM::generate_keys(const RngKeys &rng_, bool secret, bool pub, bool eval)
{
    (void)secret;
    Context &c = *ctx;
    rng = rng_;
    const size_t N = c.N;
    const int K = c.K;
    const dim3 gu((unsigned)(N / (kRngCoefs * kVmThreads)), (unsigned)K), gs((unsigned)(N / (kRngCoefs * kVmThreads))),
        g2((unsigned)(N / (2 * kVmThreads)), (unsigned)K);
    keys.sk = dalloc((size_t)K * N);
    if (secret_weight > 0) {
        // Sparse ternary secret with exactly `secret_weight` non-zero coefficients (what bootstrappable parameter sets use: the ModRaise
        // overflow I of ckks_boot.py is ~ sqrt(h / 12) wide).  Positions and signs from the secret ChaCha20 key, drawn on the host.
        if ((size_t)secret_weight > N / 2) {
            fprintf(stderr, "[dacapo_amd] option secret_hw=%d is not sparse for N = %zu\n", secret_weight, N);
            abort();
        }
        std::vector<int8_t> coef(N, 0);
        u64 w[8];
        int placed = 0;
        for (u64 blk = 0; placed < secret_weight; blk++) {
            rng_words8(rng.secret, 0, blk, 0, 1, RNG_SK, w);
            for (int e = 0; e < 8 && placed < secret_weight; e++) {
                const size_t idx = (size_t)((w[e] >> 8) % N); // N is a power of two: unbiased
                if (coef[idx]) continue;
                coef[idx] = (w[e] & 1) ? 1 : -1;
                placed++;
            }
        }
        std::vector<u64> h((size_t)K * N);
        for (int i = 0; i < K; i++)
            for (size_t k = 0; k < N; k++) h[(size_t)i * N + k] = coef[k] == 0 ? 0 : coef[k] > 0 ? 1 : c.primes[(size_t)i] - 1;
        DC_HIP_CHECK(hipMemcpyAsync(keys.sk, h.data(), h.size() * 8, hipMemcpyHostToDevice, S()));
        DC_HIP_CHECK(hipStreamSynchronize(S()));
    } else
        DC_LAUNCH(sample_small_kernel, gs, dim3(kVmThreads), 0, S(), keys.sk, N, K, 0, rng.secret, (u64)0, (u32)RNG_SK, c.d_mods);
    launch_ntt(c, false, keys.sk, (long)N, K, nullptr, 0, 0, S());
    if (pub) {
        keys.pk = dalloc((size_t)2 * K * N);
        u64 *c0 = keys.pk, *c1 = keys.pk + (size_t)K * N;
        DC_LAUNCH(sample_uniform_kernel, gu, dim3(kVmThreads), 0, S(), c1, N, rng.pub, (u64)0, (u32)RNG_PK_A, c.d_mods);
        DC_LAUNCH(sample_small_kernel, gs, dim3(kVmThreads), 0, S(), c0, N, K, 1, rng.secret, (u64)0, (u32)RNG_PK_E, c.d_mods);
        launch_ntt(c, false, c0, (long)N, K, nullptr, 0, 0, S());
        DC_LAUNCH(ezs_final_kernel, g2, dim3(kVmThreads), 0, S(), c0, c1, keys.sk, (const u64 *)nullptr, -1, -1, (const u64 *)nullptr, N,
                           c.d_mods);
    }
    if (eval) {
        u64 *sk2 = W().ct_tmp;
        launch_ew(c, EwOp::Mul, CtView{ sk2, 0 }, CtView{ keys.sk, 0 }, CtView{ keys.sk, 0 }, 1, 1, K, S());
        keys.relin = dalloc(key_elems());
        gen_kswitch_key(keys.relin, sk2, 8);
        // GaloisTool::get_elts_all(): 2N-1, then 3^(2^i), 3^-(2^i), i < logN-1
        const u64 m = 2 * (u64)N;
        add_galois_key((u32)(m - 1));
        u64 pos = 3, neg = 1;
        for (u64 x = 1; x < m; x += 2)
            if (((x * 3) & (m - 1)) == 1) {
                neg = x;
                break;
            }
        for (int i = 0; i < c.logN - 1; i++) {
            add_galois_key((u32)pos);
            add_galois_key((u32)neg);
            pos = (pos * pos) & (m - 1);
            neg = (neg * neg) & (m - 1);
        }
    }
    DC_HIP_CHECK(hipStreamSynchronize(S()));
}

// ---- key files: SEAL 4.0 serialization (seal_serial.hpp), the five files of SEAL_HEVM.cpp:55-88 / :91-180 -------------
static std::string join(const std::string &dir, const char *name)
{
    return (!dir.empty() && dir.back() == '/') ? dir + name : dir + "/" + name;
}

static std::vector<u64> from_dev(const u64 *d, size_t elems)
{
    std::vector<u64> h(elems);
    DC_HIP_CHECK(hipMemcpy(h.data(), d, elems * 8, hipMemcpyDeviceToHost));
    return h;
}
static u64 *to_dev(const u64 *h, size_t elems)
{ // `h` may be unaligned (it points into a file image): hipMemcpy takes any byte pointer
    u64 *d = dalloc(elems);
    DC_HIP_CHECK(hipMemcpy(d, h, elems * 8, hipMemcpyHostToDevice));
    return d;
}

sealio::ParmsId HEVM::parms_id_at(int limbs) const { return sealio::parms_id(ctx->N, ctx->primes.data(), (size_t)limbs); }

// KSwitchKeys::save_members: keys_[index][digit] = PublicKey; present[index] = device key [K-1][2][K][N] or absent
static void put_kswitch_keys(sealio::Writer &w, const Context &c, const sealio::ParmsId &key_id, size_t dim1,
                             const std::map<size_t, const u64 *> &present)
{
    const size_t per_digit = (size_t)2 * c.K * c.N;
    w.buf.reserve(w.buf.size() + 64 + dim1 * 8 + present.size() * (size_t)c.key_digits() * (per_digit * 8 + 128));
    w.put(key_id);
    w.put<uint64_t>(dim1);
    sealio::CtHeader h;
    h.id = key_id, h.is_ntt = true, h.size = 2, h.N = c.N, h.limbs = (uint64_t)c.K, h.correction_factor = 1, h.scale = 1.0;
    for (size_t index = 0; index < dim1; index++) {
        auto it = present.find(index);
        if (it == present.end()) {
            w.put<uint64_t>(0);
            continue;
        }
        w.put<uint64_t>((uint64_t)c.key_digits());
        const std::vector<u64> key = from_dev(it->second, (size_t)c.key_digits() * per_digit);
        for (int j = 0; j < c.key_digits(); j++) { // PublicKey::save = Ciphertext::save (own header, compr none)
            sealio::Writer one;
            one.buf.reserve(per_digit * 8 + 128);
            put_ciphertext(one, h, key.data() + (size_t)j * per_digit);
            w.put_header(one.buf.size());
            w.put_bytes(one.buf.data(), one.buf.size());
        }
    }
}

void HEVM::save_keys(const std::string &dir)
{
    const Context &c = *ctx;
    const sealio::Compr mode = sealio::compr_from_env();
    const sealio::ParmsId key_id = parms_id_at(c.K);
    {
        sealio::Writer w;
        sealio::Params p;
        p.N = c.N, p.primes = c.primes;
        put_params(w, p);
        write_object_file(join(dir, "parm.seal"), w.buf, mode);
    }
    if (keys.pk) {
        sealio::Writer w;
        sealio::CtHeader h;
        h.id = key_id, h.N = c.N, h.limbs = (uint64_t)c.K;
        put_ciphertext(w, h, from_dev(keys.pk, (size_t)2 * c.K * c.N).data());
        write_object_file(join(dir, "pub.seal"), w.buf, mode);
    }
    if (keys.sk) { // SecretKey::save = its Plaintext (NTT form over the whole chain, scale 1)
        sealio::Writer w;
        sealio::PtHeader h;
        h.id = key_id, h.coeff_count = (uint64_t)c.K * c.N, h.scale = 1.0;
        put_plaintext(w, h, from_dev(keys.sk, (size_t)c.K * c.N).data());
        write_object_file(join(dir, "sec.seal"), w.buf, mode);
    }
    if (keys.relin) { // RelinKeys: keys_[0] = the key for s^2
        sealio::Writer w;
        put_kswitch_keys(w, c, key_id, 1, { { 0, keys.relin } });
        write_object_file(join(dir, "relin.seal"), w.buf, mode);
    }
    if (!keys.galois.empty()) { // GaloisKeys: N entries, Galois element e at index (e - 1) / 2
        sealio::Writer w;
        std::map<size_t, const u64 *> present;
        for (auto &kv : keys.galois) present[(size_t)((kv.first - 1) >> 1)] = kv.second;
        put_kswitch_keys(w, c, key_id, c.N, present);
        write_object_file(join(dir, "gal.seal"), w.buf, mode);
    }
    // Grouped-digit keys (extension) use SEAL's container with another digit count and are NOT SEAL-loadable; parm.seal cannot say so, and two
    // settings with the same digit count (K = 30: 6 / 6 and 8 / 7 both give 4 digits) would load into each other's VM and compute garbage.
    // The directory therefore carries the two numbers in a sidecar that load_keys checks.
    const std::string side = join(dir, "hybrid.txt");
    if (c.hybrid()) {
        FILE *f = fopen(side.c_str(), "w");
        if (!f || fprintf(f, "ks_special=%d ks_alpha=%d\n# grouped-digit key-switching keys (dacapo_amd extension): not loadable by SEAL\n", c.ksp, c.alpha) < 0) {
            fprintf(stderr, "[dacapo_amd] cannot write %s\n", side.c_str());
            abort();
        }
        fclose(f);
    } else
        remove(side.c_str()); // (a directory rewritten in SEAL mode must not keep an older sidecar)
}

// one PublicKey of a key-switch key, or pub.seal: checks it against the context and returns its [2][K][N] limbs
static const u64 *get_key_ciphertext(sealio::Reader &members, const Context &c, const sealio::ParmsId &key_id)
{
    const u64 *data = nullptr;
    const sealio::CtHeader h = get_ciphertext(members, data);
    if (h.size != 2 || h.N != c.N || h.limbs != (uint64_t)c.K || !h.is_ntt) members.fail("key polynomial dimensions do not match parm.seal");
    if (h.id != key_id) members.fail("parms_id is not the key-level id of parm.seal (key generated under other parameters?)");
    return data;
}

// KSwitchKeys::load_members: calls sink(index, digits [K-1][2][K][N] on the device) for every non-empty entry
template <class Sink>
static void get_kswitch_keys(sealio::Reader &r, const Context &c, const sealio::ParmsId &key_id, Sink sink)
{
    if (r.get<sealio::ParmsId>() != key_id) r.fail("parms_id is not the key-level id of parm.seal");
    const uint64_t dim1 = r.get<uint64_t>();
    if (dim1 > (1ull << 20)) r.fail("implausible key count");
    const size_t per_digit = (size_t)2 * c.K * c.N;
    for (uint64_t index = 0; index < dim1; index++) {
        const uint64_t dim2 = r.get<uint64_t>();
        if (dim2 == 0) continue;
        if (dim2 != (uint64_t)c.key_digits()) r.fail("decomposition digit count differs from this context's (coeff_modulus_size - 1 in SEAL's scheme)");
        u64 *key = dalloc((size_t)c.key_digits() * per_digit);
        for (uint64_t j = 0; j < dim2; j++) {
            std::vector<uint8_t> owned;
            sealio::Reader m = open_object(r, owned);
            const u64 *data = get_key_ciphertext(m, c, key_id);
            DC_HIP_CHECK(hipMemcpy(key + j * per_digit, data, per_digit * 8, hipMemcpyHostToDevice));
        }
        sink((size_t)index, key);
    }
}

void HEVM::load_keys(const std::string &dir, bool need_secret, bool need_public, bool need_eval)
{
    auto open_file = [&](const char *name, std::vector<uint8_t> &file, std::vector<uint8_t> &owned) {
        const std::string p = join(dir, name);
        file = sealio::read_file(p);
        sealio::Reader outer(file.data(), file.size(), p);
        return open_object(outer, owned);
    };
    {
        std::vector<uint8_t> file, owned;
        sealio::Reader m = open_file("parm.seal", file, owned);
        const sealio::Params p = get_params(m);
        if (p.scheme != sealio::kSchemeCkks) m.fail("scheme is not CKKS");
        int logN = 0;
        while (((uint64_t)1 << logN) < p.N) logN++;
        for (u64 q : p.primes)
            {
                int qb = 0;
                while ((q >> qb) != 0) qb++;
                if (!prime_shape_ok(q, qb))
                    m.fail("coefficient modulus outside this backend's range: every prime must be 2^b - d with 45 <= b <= 60, d < 2^28 and d 2^(64-b) < q "
                           "(CoeffModulus::Create(N, {60, ...}) of SEAL_HEVM.cpp:48-53 yields b = 60)");
            }
        int dir_ksp = 0, dir_alpha = 0;
        if (FILE *f = fopen(join(dir, "hybrid.txt").c_str(), "r")) {
            if (fscanf(f, "ks_special=%d ks_alpha=%d", &dir_ksp, &dir_alpha) != 2 || dir_ksp < 1 || dir_alpha < 1 || dir_alpha > dir_ksp) {
                fclose(f);
                fprintf(stderr, "[dacapo_amd] %s/hybrid.txt is malformed (expected \"ks_special=<k> ks_alpha=<a>\" with 1 <= a <= k)\n", dir.c_str());
                abort();
            }
            fclose(f);
        }
        init_context(logN, (int)p.primes.size(), p.primes.data(), dir_ksp, dir_alpha);
        if (ctx->hybrid() && dir_ksp == 0) {
            fprintf(stderr, "[dacapo_amd] options ask for grouped-digit key switching (ks_special = %d, ks_alpha = %d) but %s has no hybrid.txt: it holds "
                            "SEAL-format keys, or grouped-digit keys written before the sidecar existed (round 3).  For the latter, regenerate the "
                            "keys (create_context) or write \"ks_special=%d ks_alpha=%d\" into %s/hybrid.txt by hand\n",
                    ctx->ksp, ctx->alpha, dir.c_str(), ctx->ksp, ctx->alpha, dir.c_str());
            abort();
        }
    }
    const Context &c = *ctx;
    const sealio::ParmsId key_id = parms_id_at(c.K);
    if (need_public) {
        std::vector<uint8_t> file, owned;
        sealio::Reader m = open_file("pub.seal", file, owned);
        keys.pk = to_dev(get_key_ciphertext(m, c, key_id), (size_t)2 * c.K * c.N);
    }
    if (need_secret) {
        std::vector<uint8_t> file, owned;
        sealio::Reader m = open_file("sec.seal", file, owned);
        const u64 *data = nullptr;
        const sealio::PtHeader h = get_plaintext(m, data);
        if (h.coeff_count != (uint64_t)c.K * c.N || h.id != key_id) m.fail("secret key does not match parm.seal");
        keys.sk = to_dev(data, (size_t)c.K * c.N);
    }
    if (need_eval) {
        {
            std::vector<uint8_t> file, owned;
            sealio::Reader m = open_file("relin.seal", file, owned);
            get_kswitch_keys(m, c, key_id, [&](size_t index, u64 *key) {
                if (index == 0)
                    keys.relin = key;
                else
                    (void)vm_free(key); // keys for higher powers of s: the HEVM path never multiplies without relinearising
            });
            if (!keys.relin) m.fail("no relinearisation key for s^2");
        }
        std::vector<uint8_t> file, owned;
        sealio::Reader m = open_file("gal.seal", file, owned);
        get_kswitch_keys(m, c, key_id, [&](size_t index, u64 *key) { keys.galois[(u32)(2 * index + 1)] = key; });
    }
    rng = rng_keys_from_os(); // encryption randomness of this VM: fresh from the OS, unrelated to whatever generated the loaded keys
}

// seal::Ciphertext::save / load of a cipher register (what getCtxt's seal::Ciphertext* is for in the reference,
// SEAL_HEVM.cpp:463-473 "use this to implement communication"): [2][level][N] limbs, the parms_id of that level, the scale
void HEVM::save_ctxt(size_t r, const std::string &path)
{
    const Context &c = *ctx;
    hevm_ctxt &ct = reg(r);
    if (ct.level < 1) {
        fprintf(stderr, "[dacapo_amd] save: cipher register %zu is empty\n", r);
        abort();
    }
    DC_HIP_CHECK(hipStreamSynchronize(S()));
    const size_t ln = (size_t)ct.level * c.N;
    std::vector<u64> h(2 * ln);
    for (int p = 0; p < 2; p++) DC_HIP_CHECK(hipMemcpy(h.data() + (size_t)p * ln, ct.data + (size_t)p * ct.poly_stride, ln * 8, hipMemcpyDeviceToHost));
    sealio::Writer w;
    sealio::CtHeader hd;
    hd.id = parms_id_at(ct.level), hd.N = c.N, hd.limbs = (uint64_t)ct.level, hd.scale = ct.scale;
    put_ciphertext(w, hd, h.data());
    write_object_file(path, w.buf, sealio::compr_from_env());
}

void HEVM::load_ctxt(size_t r, const std::string &path)
{
    const Context &c = *ctx;
    const std::vector<uint8_t> file = sealio::read_file(path);
    std::vector<uint8_t> owned;
    sealio::Reader outer(file.data(), file.size(), path);
    sealio::Reader m = open_object(outer, owned);
    const u64 *data = nullptr;
    const sealio::CtHeader h = get_ciphertext(m, data);
    if (h.size != 2 || h.N != c.N || h.limbs < 1 || h.limbs > (uint64_t)c.max_level() || !h.is_ntt) m.fail("not a size-2 NTT-form ciphertext of this context");
    if (h.id != parms_id_at((int)h.limbs)) m.fail("parms_id does not name a level of this context's modulus chain");
    hevm_ctxt &ct = reg(r);
    reg_base[r] = home[r];
    ct.data = reg_base[r] + (size_t)sel * (size_t)2 * c.K * c.N;
    ct.poly_stride = (int64_t)c.K * (int64_t)c.N;
    const size_t ln = (size_t)h.limbs * c.N;
    for (int p = 0; p < 2; p++) DC_HIP_CHECK(hipMemcpy(ct.data + (size_t)p * ct.poly_stride, data + (size_t)p * ln, ln * 8, hipMemcpyHostToDevice));
    ct.level = (int32_t)h.limbs, ct.scale = h.scale;
}

// ---------------------------------------------------------------------------------------------------------
// program loading (SEAL_HEVM.cpp:182-240)
// ---------------------------------------------------------------------------------------------------------
void HEVM::load_constants(const void *data, size_t len)
{
    std::string err;
    if (!wire::parse_constants(data, len, buffer, err)) { // (wire_parse.cpp: every count held against the bytes that are there)
        fprintf(stderr, "[dacapo_amd] %s\n", err.c_str());
        abort();
    }
}

void HEVM::load_program(const void *data, size_t len, bool header_only)
{
    wire::Program pr;
    std::string err;
    if (!wire::parse_program(data, len, header_only, buffer, pr, err)) { // (wire_parse.cpp: bounds and operand validation before anything is allocated)
        fprintf(stderr, "[dacapo_amd] %s\n", err.c_str());
        abort();
    }
    header = pr.header, config = pr.config;
    arg_scale = std::move(pr.arg_scale), arg_level = std::move(pr.arg_level);
    res_scale = std::move(pr.res_scale), res_level = std::move(pr.res_level), res_dst = std::move(pr.res_dst);
    const size_t nct = pr.cipher_registers;
    if (!header_only) {
        ops = std::move(pr.ops);
        free_plains();
        plains.assign(config.num_ptxt_buffer, Plain{});
    }
    while (ciphers.size() < nct) ciphers.push_back(hevm_ctxt{ nullptr, 0, 0, 0, 1.0 });
    for (size_t i = 0; i < nct; i++) reg(i); // allocate now: nothing may call hipMalloc while run() is being captured
    plan.ready = false;
}

void HEVM::reset_res_dst()
{
    for (size_t i = 0; i < res_dst.size(); i++) res_dst[i] = i + header.arg_length;
}

hevm_ctxt &HEVM::reg(size_t i)
{
    while (i >= ciphers.size()) ciphers.push_back(hevm_ctxt{ nullptr, 0, 0, 0, 1.0 }); // deque: references stay valid
    while (i >= home.size()) home.push_back(nullptr), reg_base.push_back(nullptr);
    hevm_ctxt &r = ciphers[i];
    const size_t slice = (size_t)2 * ctx->K * ctx->N; // full capacity [2][K][N] (K limbs: encryption stages one extra prime)
    if (!home[i]) home[i] = dalloc(slice * (size_t)streams);
    if (!r.data) {
        r.poly_stride = (int64_t)ctx->K * (int64_t)ctx->N;
        reg_base[i] = home[i];
        r.data = reg_base[i] + (size_t)sel * slice;
        r.level = 0;
        r.scale = 1.0;
    }
    return r;
}

void HEVM::set_streams(int n)
{
    if (n < 1) n = 1;
    if (n == streams) return;
    DC_HIP_CHECK(hipDeviceSynchronize());
    for (u64 *&p : home)
        if (p) (void)vm_free(p), p = nullptr;
    for (u64 *p : plan.pool) (void)vm_free(p);
    plan.pool.clear();
    plan.ready = false;
    for (auto &r : ciphers) r.data = nullptr;
    streams = n;
    sel = 0;
    for (size_t i = 0; i < ciphers.size(); i++) reg(i);
}

void HEVM::select_stream(int s)
{
    if (s < 0 || s >= streams) {
        fprintf(stderr, "[dacapo_amd] stream %d outside 0..%d\n", s, streams - 1);
        abort();
    }
    sel = s;
    const size_t slice = (size_t)2 * ctx->K * ctx->N;
    for (size_t i = 0; i < ciphers.size(); i++)
        if (reg_base[i]) ciphers[i].data = reg_base[i] + (size_t)sel * slice;
}

void HEVM::bump_epoch(hipStream_t s) { DC_LAUNCH(bump_epoch_kernel, dim3(1), dim3(1), 0, s, d_epoch); }

// ---------------------------------------------------------------------------------------------------------
// encode / encrypt / decrypt (SEAL_HEVM.cpp:242-267, :439-455)
// ---------------------------------------------------------------------------------------------------------
void HEVM::encode_internal(Plain &dst, const double *src, size_t len, int level, int scale_bits)
{
    Context &c = *ctx;
    const size_t N = c.N;
    if (level < 1 || level > c.max_level()) {
        fprintf(stderr, "[dacapo_amd] encode: level %d outside 1..%d\n", level, c.max_level());
        abort();
    }
    std::vector<__int128> coeffs;
    const double scale = pow(2.0, (double)scale_bits);
    encoder->encode(src, len, scale, coeffs);
    std::vector<u64> lohi(2 * N);
    for (size_t j = 0; j < N; j++) {
        lohi[j] = (u64)(unsigned __int128)coeffs[j];
        lohi[N + j] = (u64)((unsigned __int128)coeffs[j] >> 64);
    }
    u64 *stage = W().ks_digits; // always >= 2N elements (Context::new_workspace)
    DC_HIP_CHECK(hipMemcpyAsync(stage, lohi.data(), 2 * N * 8, hipMemcpyHostToDevice, S()));
    if (dst.d && (dst.level != level || dst.arena)) {
        DC_HIP_CHECK(hipStreamSynchronize(S()));
        if (!dst.arena) (void)vm_free(dst.d);
        dst.d = nullptr, dst.arena = false;
    }
    if (!dst.d) dst.d = dalloc((size_t)level * N);
    dst.level = level;
    dst.scale = scale;
    DC_LAUNCH(lift_i128_kernel, dim3((unsigned)(N / kVmThreads), (unsigned)level), dim3(kVmThreads), 0, S(), dst.d,
                       stage, stage + N, N, c.d_mods);
    launch_ntt(c, false, dst.d, (long)N, level, nullptr, 0, 0, S());
    DC_HIP_CHECK(hipStreamSynchronize(S())); // `lohi` and the staging buffer are reused by the next call
}

void HEVM::free_plains()
{
    if (online.d_consts) (void)vm_free(online.d_consts);
    online.d_consts = nullptr, online.items.clear();
    if (dh_consts) (void)vm_free(dh_consts);
    dh_consts = nullptr, dh_items.clear();
    for (auto &pl : plains)
        if (pl.d && !pl.arena) (void)vm_free(pl.d);
    for (u64 *a : plain_arenas) (void)vm_free(a);
    plain_arenas.clear();
    for (auto &pl : plains) pl = Plain{};
}

// All opcode-0 instructions of the program at once: constants uploaded once, plaintexts grouped by level, each group
// encoded in chunks by encoder.hip (scatter, logN butterfly launches, round + reduce, NTT) straight into its arena.
void HEVM::ensure_enc_tables()
{
    const size_t N = ctx->N;
    if (!enc_tables.roots) {
        std::vector<double2> r(N);
        for (size_t k = 0; k < N; k++) r[k] = make_double2(encoder->roots()[k].real(), encoder->roots()[k].imag());
        DC_HIP_CHECK(vm_malloc(&enc_tables.roots, N * sizeof(double2)));
        DC_HIP_CHECK(hipMemcpy(enc_tables.roots, r.data(), N * sizeof(double2), hipMemcpyHostToDevice));
        DC_HIP_CHECK(vm_malloc(&enc_tables.slot_map, N * sizeof(u32)));
        DC_HIP_CHECK(hipMemcpy(enc_tables.slot_map, encoder->slot_map().data(), N * sizeof(u32), hipMemcpyHostToDevice));
    }
}

void HEVM::preprocess_device()
{
    Context &c = *ctx;
    const size_t N = c.N;
    ensure_enc_tables();
    free_plains();
    // constants referenced by the program -> one device arena
    std::vector<size_t> off(buffer.size(), (size_t)-1);
    std::vector<double> host;
    std::map<int, std::vector<std::pair<int, EncItem>>> by_level; // level -> (plain register, item)
    std::vector<long> last_write(plains.size(), -1); // a register encoded more than once keeps its last value
    for (size_t k = 0; k < ops.size(); k++)
        if (ops[k].opcode == 0 || ops[k].opcode == kOpEncodeComplex) last_write.at(ops[k].dst) = (long)k;
    for (size_t k = 0; k < ops.size(); k++) {
        const WireOp &op = ops[k];
        if ((op.opcode != 0 && op.opcode != kOpEncodeComplex) || last_write[op.dst] != (long)k) continue;
        const int level = op.rhs >> 10, scale_bits = op.rhs & 0x3FF;
        if (level < 1 || level > c.max_level()) {
            fprintf(stderr, "[dacapo_amd] encode: level %d outside 1..%d\n", level, c.max_level());
            abort();
        }
        EncItem it{ 0, 0, 0, pow(2.0, (double)scale_bits) / (double)N };
        if (op.lhs != 0xFFFF) {
            const std::vector<double> &src = buffer.at(op.lhs);
            if (src.empty()) {
                fprintf(stderr, "[dacapo_amd] encode: constant %u is empty\n", (unsigned)op.lhs);
                abort();
            }
            if (off[op.lhs] == (size_t)-1) {
                off[op.lhs] = host.size();
                host.insert(host.end(), src.begin(), src.end());
            }
            it.src_off = off[op.lhs], it.len = (u32)src.size();
            if (op.opcode == kOpEncodeComplex) {
                if (src.size() < 2 || (src.size() & 1)) {
                    fprintf(stderr, "[dacapo_amd] encode (complex): constant %u must hold real parts then imaginary parts\n", (unsigned)op.lhs);
                    abort();
                }
                it.len = (u32)(src.size() / 2), it.cplx = 1;
            }
        } else if (op.opcode == kOpEncodeComplex) {
            fprintf(stderr, "[dacapo_amd] encode (complex): needs a constant\n");
            abort();
        }
        Plain &pl = plains.at(op.dst);
        pl.level = level, pl.scale = pow(2.0, (double)scale_bits), pl.arena = true;
        by_level[level].push_back({ (int)op.dst, it });
        if (online_encode) online.items[(int)op.dst] = it;
        if (double_hoist) dh_items[(int)op.dst] = it;
    }
    if (by_level.empty()) return;
    double *d_consts = nullptr;
    DC_HIP_CHECK(vm_malloc(&d_consts, std::max<size_t>(host.size(), 1) * sizeof(double)));
    if (!host.empty()) DC_HIP_CHECK(hipMemcpy(d_consts, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
    if (online_encode) {
        // On-line encode (the HEaaN runtime's mode, HEAAN_HEVM.cpp:266-281,353-363): only the constants stay resident, as doubles;
        // the plan encodes every plaintext register into a recycled window right before the wave that first reads it
        // (build_plan: Plan::enc_groups).  Registers get their addresses there.
        online.d_consts = d_consts, online.const_bytes = host.size() * sizeof(double);
        return;
    }
    const int chunk = 1024;
    double2 *scratch = nullptr;
    EncItem *d_items = nullptr;
    int *d_overflow = nullptr;
    DC_HIP_CHECK(vm_malloc(&scratch, (size_t)chunk * N * sizeof(double2)));
    DC_HIP_CHECK(vm_malloc(&d_items, (size_t)chunk * sizeof(EncItem)));
    DC_HIP_CHECK(vm_malloc(&d_overflow, sizeof(int)));
    DC_HIP_CHECK(hipMemset(d_overflow, 0, sizeof(int)));
    for (auto &kv : by_level) {
        const int level = kv.first;
        const size_t P = kv.second.size();
        u64 *arena = dalloc(P * (size_t)level * N);
        plain_arenas.push_back(arena);
        std::vector<EncItem> items(P);
        for (size_t k = 0; k < P; k++) {
            items[k] = kv.second[k].second;
            plains.at((size_t)kv.second[k].first).d = arena + k * (size_t)level * N;
        }
        for (size_t k = 0; k < P; k += (size_t)chunk) {
            const int cnt = (int)std::min<size_t>((size_t)chunk, P - k);
            DC_HIP_CHECK(hipMemcpyAsync(d_items, items.data() + k, (size_t)cnt * sizeof(EncItem), hipMemcpyHostToDevice, S()));
            enc_batch(c, enc_tables, d_consts, d_items, cnt, level, scratch, arena + k * (size_t)level * N, d_overflow, S());
            DC_HIP_CHECK(hipStreamSynchronize(S())); // d_items is reused by the next chunk
        }
    }
    int overflow = 0;
    DC_HIP_CHECK(hipMemcpy(&overflow, d_overflow, sizeof(int), hipMemcpyDeviceToHost));
    if (double_hoist) // (build_plan encodes special-prime limbs from the same constants: they stay until the plaintexts go)
        dh_consts = d_consts;
    else
        (void)vm_free(d_consts);
    (void)vm_free(scratch), (void)vm_free(d_items), (void)vm_free(d_overflow);
    if (overflow) {
        fprintf(stderr, "[dacapo_amd] encode: coefficient does not fit 120 bits (scale too large)\n");
        abort();
    }
}

// option hyb_double_hoist: the limbs over the chain's SPECIAL primes of the plaintext registers the plan multiplies rotations by inside its lazy sums
// (plan_exec.hip section 2b).  The same encoding as preprocess_device's -- same items, same constants, same FFT, hence the same integer
// coefficients -- reduced into primes max_level ... max_level + ksp - 1; registers that already have them are skipped.
void HEVM::ensure_special_limbs(const std::vector<int> &plain_regs)
{
    Context &c = *ctx;
    const size_t N = c.N;
    std::vector<int> todo;
    for (int r : plain_regs)
        if (!plains.at((size_t)r).dsp) todo.push_back(r);
    std::sort(todo.begin(), todo.end());
    todo.erase(std::unique(todo.begin(), todo.end()), todo.end());
    if (todo.empty()) return;
    if (!dh_consts) {
        fprintf(stderr, "[dacapo_amd] double hoisting: the program's constants are gone (option hyb_double_hoist was off, or host_encoder / online_encode is on, when the program was preprocessed)\n");
        abort();
    }
    ensure_enc_tables();
    const int ksp = c.ksp, chunk = 256;
    u64 *arena = dalloc(todo.size() * (size_t)ksp * N);
    plain_arenas.push_back(arena);
    double2 *scratch = nullptr;
    EncItem *d_items = nullptr;
    int *d_overflow = nullptr;
    DC_HIP_CHECK(vm_malloc(&scratch, (size_t)chunk * N * sizeof(double2)));
    DC_HIP_CHECK(vm_malloc(&d_items, (size_t)chunk * sizeof(EncItem)));
    DC_HIP_CHECK(vm_malloc(&d_overflow, sizeof(int)));
    DC_HIP_CHECK(hipMemset(d_overflow, 0, sizeof(int)));
    std::vector<EncItem> items(todo.size());
    for (size_t k = 0; k < todo.size(); k++) {
        items[k] = dh_items.at(todo[k]);
        plains.at((size_t)todo[k]).dsp = arena + k * (size_t)ksp * N;
    }
    for (size_t k = 0; k < todo.size(); k += (size_t)chunk) {
        const int cnt = (int)std::min<size_t>((size_t)chunk, todo.size() - k);
        DC_HIP_CHECK(hipMemcpyAsync(d_items, items.data() + k, (size_t)cnt * sizeof(EncItem), hipMemcpyHostToDevice, S()));
        enc_batch(c, enc_tables, dh_consts, d_items, cnt, ksp, scratch, arena + k * (size_t)ksp * N, d_overflow, S(), c.max_level());
        DC_HIP_CHECK(hipStreamSynchronize(S()));
    }
    int overflow = 0;
    DC_HIP_CHECK(hipMemcpy(&overflow, d_overflow, sizeof(int), hipMemcpyDeviceToHost));
    (void)vm_free(scratch), (void)vm_free(d_items), (void)vm_free(d_overflow);
    if (overflow) {
        fprintf(stderr, "[dacapo_amd] encode: coefficient does not fit 120 bits (scale too large)\n");
        abort();
    }
}

void HEVM::preprocess()
{
    plan.ready = false; // plaintext registers may move: item tables of an existing plan would dangle
    if (!host_encoder)
        preprocess_device();
    else {
        const std::vector<double> identity(1, 1.0); // tiled to all ones, like the reference's identity vector
        for (const WireOp &op : ops)
            if (op.opcode == kOpEncodeComplex) {
                fprintf(stderr, "[dacapo_amd] complex plaintexts (extension opcode 16) need the device encoder\n");
                abort();
            } else if (op.opcode == 0) {
                const std::vector<double> &src = (op.lhs == 0xFFFF) ? identity : buffer.at(op.lhs);
                encode_internal(plains.at(op.dst), src.data(), src.size(), op.rhs >> 10, op.rhs & 0x3FF);
            }
    }
    // The reference times run() alone (examples/tests/ResNet.py:109-111) after an untimed preprocess(): the execution plan
    // of the loaded program is part of the preparation, not of the run.  (A VM without evaluation keys cannot run anyway.)
    if (use_plan && !debug && keys.relin && !keys.galois.empty()) build_plan();
}

// Encryptor::encrypt at the plaintext's level: zero-encryption under pk with one extra prime, divide-and-round by it,
// add the plaintext.
void HEVM::encrypt_plain(hevm_ctxt &dst, const Plain &pt)
{
    Context &c = *ctx;
    const size_t N = c.N;
    const int ell = pt.level, cnt = ell + 1;
    if (!keys.pk) {
        fprintf(stderr, "[dacapo_amd] encrypt: this VM has no public key\n");
        abort();
    }
    if (test_zero_enc) { // TEST HOOK: (plaintext, 0)
        DC_HIP_CHECK(hipMemcpyAsync(dst.data, pt.d, (size_t)ell * N * 8, hipMemcpyDeviceToDevice, S()));
        DC_HIP_CHECK(hipMemsetAsync(dst.data + dst.poly_stride, 0, (size_t)ell * N * 8, S()));
        dst.level = ell, dst.scale = pt.scale;
        return;
    }
    u64 *ue = W().ks_ext; // [3][cnt][N]: u, e0, e1
    const CtView tmp{ dst.data, (long)dst.poly_stride };
    // one object per encryption of this VM's lifetime (opcode-10 items of a plan use the range above 2^32)
    DC_LAUNCH(sample_enc_batch_kernel, dim3((unsigned)(N / (kRngCoefs * kVmThreads)), 3), dim3(kVmThreads), 0, S(), ue, N, cnt,
                       rng.secret, enc_counter++, c.d_mods, d_epoch);
    launch_ntt(c, false, ue, (long)N, 3 * cnt, nullptr, 0, cnt, S());
    DC_LAUNCH(pk_encrypt_kernel, dim3((unsigned)(N / (2 * kVmThreads)), (unsigned)cnt, 2), dim3(kVmThreads), 0, S(),
                       tmp.p, tmp.poly_stride, keys.pk, (long)c.K * (long)N, ue, ue + (size_t)cnt * N, (long)cnt * (long)N, N, c.d_mods);
    rescale_fused(c, W(), tmp, tmp, cnt, pt.d, S()); // divide-and-round by the extra prime, then + plaintext on c0
    dst.level = ell;
    dst.scale = pt.scale;
}

void HEVM::encrypt(int64_t i, const double *dat, int len)
{
    Plain pt;
    encode_internal(pt, dat, (size_t)len, (int)arg_level.at((size_t)i), (int)arg_scale.at((size_t)i));
    hevm_ctxt &r = reg((size_t)i);
    reg_base[(size_t)i] = home[(size_t)i]; // program inputs always live in the register's own block (a plan may have re-pointed it)
    r.data = reg_base[(size_t)i] + (size_t)sel * (size_t)2 * ctx->K * ctx->N;
    r.poly_stride = (int64_t)ctx->K * (int64_t)ctx->N;
    encrypt_plain(r, pt);
    DC_HIP_CHECK(hipStreamSynchronize(S()));
    (void)vm_free(pt.d);
}

void HEVM::decrypt(int64_t i, double *out)
{
    Context &c = *ctx;
    const size_t N = c.N;
    hevm_ctxt &ct = reg((size_t)i);
    if (!keys.sk || ct.level < 1) {
        fprintf(stderr, "[dacapo_amd] decrypt: no secret key or empty register %lld\n", (long long)i);
        abort();
    }
    u64 *pt = W().ks_tmp;
    // Above 16 primes the modulus leaves a double's range (2^960) and the device CRT takes the first 16 limbs only: the plaintext polynomial
    // m = c0 + c1 s mod Q has |coefficients| < Q_16 / 2 for anything that decodes to finite doubles at all (an integer beyond 2^959 divided by
    // any admissible scale is not a CKKS message but overflowed noise), and then its centred representative mod Q_16 is the one mod Q_ell.
    // Limbs are independent, so decrypting and inverse-transforming only those 16 changes nothing about them.
    const int ell = !host_encoder && ct.level > 16 ? 16 : ct.level;
    DC_LAUNCH(decrypt_kernel, dim3((unsigned)(N / (2 * kVmThreads)), (unsigned)ell), dim3(kVmThreads), 0, S(), pt,
                       view(ct), keys.sk, N, c.d_mods);
    launch_ntt(c, true, pt, (long)N, ell, nullptr, 0, 0, S());
    if (!host_encoder) { // compose + forward special FFT + slot gather on the device; only the N/2 slot values cross PCIe
        ensure_enc_tables();
        const CrtTables &tb = crt_tables(ell);
        const CrtDev cd{ tb.inv, tb.mmod, tb.hmod, tb.hdig, tb.mdbl };
        double2 *v = reinterpret_cast<double2 *>(W().ks_ext); // >= 3K limbs = 3K*N u64 >= 2N doubles
        double *slots = reinterpret_cast<double *>(W().ks_acc);
        DC_LAUNCH(dec_crt_kernel, dim3((unsigned)(N / kVmThreads)), dim3(kVmThreads), 0, S(), v, pt, ell, N, c.d_mods, cd, 1.0 / ct.scale);
        dec_fft(c, enc_tables, v, slots, S());
        DC_HIP_CHECK(hipMemcpyAsync(out, slots, (N / 2) * sizeof(double), hipMemcpyDeviceToHost, S()));
        DC_HIP_CHECK(hipStreamSynchronize(S()));
        return;
    }
    std::vector<u64> coef((size_t)ell * N);
    DC_HIP_CHECK(hipMemcpyAsync(coef.data(), pt, coef.size() * 8, hipMemcpyDeviceToHost, S()));
    DC_HIP_CHECK(hipStreamSynchronize(S()));
    // CRT-compose each coefficient (Garner), centre it, divide by the scale  [CKKSEncoder::decode_internal]
    const int W = ell;
    std::vector<u64> M((size_t)(ell + 1) * W, 0); // M[k] = q_0 ... q_{k-1}
    M[0] = 1;
    auto mul_add = [&](u64 *acc, const u64 *a, u64 b) {
        u128 carry = 0;
        for (int w = 0; w < W; w++) {
            u128 t = (u128)a[w] * b + acc[w] + (u64)carry;
            acc[w] = (u64)t;
            carry = t >> 64;
        }
    };
    for (int k = 1; k <= ell; k++) mul_add(&M[(size_t)k * W], &M[(size_t)(k - 1) * W], c.primes[k - 1]);
    std::vector<std::vector<u64>> Mmod(ell, std::vector<u64>(ell, 0));
    std::vector<u64> inv(ell);
    for (int k = 0; k < ell; k++) {
        const u64 qk = c.primes[k];
        u64 acc = 1;
        for (int j = 0; j <= k; j++) {
            Mmod[j][k] = acc;
            if (j < k) acc = h_mulmod(acc, c.primes[j] % qk, qk);
        }
        inv[k] = h_invmod(Mmod[k][k], qk);
    }
    const u64 *Q = &M[(size_t)ell * W];
    std::vector<u64> half(W);
    {
        std::vector<u64> t(W);
        u64 carry = 1;
        for (int w = 0; w < W; w++) {
            t[w] = Q[w] + carry;
            carry = (carry && t[w] == 0) ? 1 : 0;
        }
        for (int w = 0; w < W; w++) half[w] = (t[w] >> 1) | ((w + 1 < W ? t[w + 1] : carry) << 63);
    }
    std::vector<std::complex<double>> res(N);
    const double inv_scale = 1.0 / ct.scale;
    std::vector<u64> v(ell), X(W);
    for (size_t n = 0; n < N; n++) {
        for (int k = 0; k < ell; k++) {
            const u64 qk = c.primes[k];
            u64 s = 0;
            for (int j = 0; j < k; j++) s = (s + h_mulmod(v[j] % qk, Mmod[j][k], qk)) % qk;
            v[k] = h_mulmod((coef[(size_t)k * N + n] + qk - s) % qk, inv[k], qk);
        }
        std::fill(X.begin(), X.end(), 0);
        for (int k = 0; k < ell; k++) mul_add(X.data(), &M[(size_t)k * W], v[k]);
        bool upper = true; // X >= (Q+1)/2 ?
        for (int w = W - 1; w >= 0; w--)
            if (X[w] != half[w]) {
                upper = X[w] > half[w];
                break;
            }
        double r = 0.0, sc = inv_scale;
        for (int w = 0; w < W; w++, sc *= 0x1p64) {
            if (!upper)
                r += X[w] ? (double)X[w] * sc : 0.0;
            else if (X[w] > Q[w])
                r += (double)(X[w] - Q[w]) * sc;
            else if (X[w] < Q[w])
                r -= (double)(Q[w] - X[w]) * sc;
        }
        res[n] = r;
    }
    encoder->decode(res, out);
}

// ---------------------------------------------------------------------------------------------------------
// opcode handlers (SEAL_HEVM.cpp:268-334) and dispatch (:336-401)
// ---------------------------------------------------------------------------------------------------------
std::vector<u32> HEVM::rotate_hops(int steps) const
{ // Evaluator::rotate_internal: direct key if present, else NAF digits (least significant first)
    std::vector<u32> hops;
    if (steps == 0) return hops;
    const int n = (int)ctx->N;
    const int pos = steps < 0 ? -steps : steps;
    if (pos >= (n >> 1)) {
        fprintf(stderr, "[dacapo_amd] rotate: step count too large (%d)\n", steps);
        abort();
    }
    u64 e = steps < 0 ? (u64)((n >> 1) - pos) : (u64)pos, elt = 1, g = 3;
    const u64 m = 2 * (u64)n;
    for (; e; e >>= 1) {
        if (e & 1) elt = (elt * g) & (m - 1);
        g = (g * g) & (m - 1);
    }
    if (keys.galois.count((u32)elt)) {
        hops.push_back((u32)elt);
        return hops;
    }
    if (rot_compose) { // option rot_compose: the fewest hops over the keys this VM holds (see compose_rotation)
        const std::vector<int> parts = compose_rotation(steps);
        if (!parts.empty()) {
            for (int p : parts) {
                const std::vector<u32> h = rotate_hops(p); // (each part has a direct key)
                hops.insert(hops.end(), h.begin(), h.end());
            }
            return hops;
        }
    }
    std::vector<int> naf;
    {
        int value = pos;
        const bool sign = steps < 0;
        for (int i = 0; value; i++) {
            const int zi = (value & 1) ? 2 - (value & 3) : 0;
            value = (value - zi) >> 1;
            if (zi) naf.push_back((sign ? -zi : zi) * (1 << i));
        }
    }
    if (naf.size() == 1) {
        fprintf(stderr, "[dacapo_amd] rotate: Galois key not present for step %d\n", steps);
        abort();
    }
    for (int s : naf)
        if (std::abs(s) != (n >> 1)) {
            std::vector<u32> h = rotate_hops(s);
            hops.insert(hops.end(), h.begin(), h.end());
        }
    return hops;
}

// EXTENSION (option rot_compose = 1; off by default: SEAL's rotate_internal only ever composes from the power-of-two keys): a bounded key
// set serving every offset, the way the reference's HEaaN runtime serves every rotation of a program from its 49 left-rotation keys
// (HEAAN_HEVM.cpp:58-64,124-126).  A rotation without a direct key becomes the SHORTEST sum of offsets that have one -- two parts if any
// pair fits, else three -- found in a fixed order (candidates ascending by |offset|, positive before negative), so that the oracle's
// restatement (oracle/oracle.py rotate_hops) picks the same parts in the same order and limbs stay comparable.  Empty: no sum of <= 3.
std::vector<int> HEVM::compose_rotation(int steps) const
{
    const int slots = (int)(ctx->N >> 1);
    auto norm = [&](long v) {
        v %= slots;
        if (v > slots / 2) v -= slots;
        if (v <= -slots / 2) v += slots;
        return (int)v;
    };
    // (the cache is keyed on the SET of Galois elements held, not on its size: a key set replaced by another of equal size has other offsets)
    size_t stamp = keys.galois.size();
    for (const auto &kv : keys.galois) stamp = stamp * 0x9E3779B97F4A7C15ull + kv.first;
    if (rot_offsets_epoch != stamp) { // the offsets that have a key, from the Galois elements 3^k
        rot_offsets.clear();
        std::map<u32, int> step_of;
        const u64 m = 2 * (u64)ctx->N;
        u64 e = 1;
        for (int k = 0; k < slots; k++, e = (e * 3) & (m - 1))
            if (keys.galois.count((u32)e)) step_of[(u32)e] = norm(k);
        for (auto &kv : step_of)
            if (kv.second != 0) rot_offsets.push_back(kv.second);
        std::sort(rot_offsets.begin(), rot_offsets.end(), [](int a, int b) { return std::abs(a) != std::abs(b) ? std::abs(a) < std::abs(b) : a > b; });
        rot_offset_set.clear();
        rot_offset_set.insert(rot_offsets.begin(), rot_offsets.end());
        rot_offsets_epoch = stamp;
    }
    const int t = norm(steps);
    for (int a : rot_offsets)
        if (rot_offset_set.count(norm((long)t - a))) return { a, norm((long)t - a) };
    for (int a : rot_offsets)
        for (int b : rot_offsets)
            if (rot_offset_set.count(norm((long)t - a - b))) return { a, b, norm((long)t - a - b) };
    return {};
}

#define ks_ntts(ell) ks_ntt_count(*ctx, (ell))

void HEVM::op_rotate(int dst, int src, int offset)
{
    hevm_ctxt &s = reg(src);
    hevm_ctxt &d = reg(dst);
    if (debug) std::cout << std::log2(s.scale) << std::endl;
    const std::vector<u32> hops = rotate_hops(offset);
    const int ell = s.level;
    const hevm_ctxt *cur = &s;
    for (u32 elt : hops) {
        rotate_hop(*ctx, W(), view(d), view(*cur), elt, keys.galois.at(elt), ell, S());
        cur = &d;
        n_keyswitch++, n_ntt += ks_ntts(ell);
    }
    if (hops.empty() && dst != src) launch_ew(*ctx, EwOp::Copy, view(d), view(s), view(s), 2, 2, ell, S());
    d.level = ell, d.scale = s.scale;
}
void HEVM::op_negate(int dst, int src)
{
    hevm_ctxt &s = reg(src);
    hevm_ctxt &d = reg(dst);
    if (debug) std::cout << std::log2(s.scale) << std::endl;
    launch_ew(*ctx, EwOp::Neg, view(d), view(s), view(s), 2, 2, s.level, S());
    d.level = s.level, d.scale = s.scale;
}
void HEVM::op_rescale(int dst, int src)
{
    hevm_ctxt &s = reg(src);
    hevm_ctxt &d = reg(dst);
    if (debug) std::cout << std::log2(s.scale) << std::endl;
    const int ell = s.level;
    if (ell < 2) {
        fprintf(stderr, "[dacapo_amd] rescale: end of modulus switching chain reached\n");
        abort();
    }
    rescale(*ctx, W(), view(d), view(s), ell, S());
    n_ntt += 2 * ell;
    d.scale = s.scale / (double)ctx->primes[ell - 1];
    d.level = ell - 1;
}
void HEVM::op_modswitch(int dst, int src, int down)
{
    hevm_ctxt &s = reg(src);
    hevm_ctxt &d = reg(dst);
    if (debug) std::cout << std::log2(s.scale) << std::endl;
    if (down <= 0) return; // the reference leaves dst untouched (SEAL_HEVM.cpp:288)
    const int ell = s.level - down;
    if (ell < 1) {
        fprintf(stderr, "[dacapo_amd] modswitch: end of modulus switching chain reached\n");
        abort();
    }
    if (dst != src) launch_ew(*ctx, EwOp::Copy, view(d), view(s), view(s), 2, 2, ell, S());
    d.level = ell, d.scale = s.scale;
}
void HEVM::op_addcc(int dst, int lhs, int rhs)
{
    hevm_ctxt &a = reg(lhs);
    hevm_ctxt &b = reg(rhs);
    hevm_ctxt &d = reg(dst);
    if (debug) std::cout << std::log2(a.scale) << std::log2(b.scale) << std::endl;
    a.scale = b.scale; // SEAL_HEVM.cpp:301
    if (a.level != b.level) {
        fprintf(stderr, "[dacapo_amd] addcc: level mismatch %d vs %d\n", a.level, b.level);
        abort();
    }
    launch_ew(*ctx, EwOp::Add, view(d), view(a), view(b), 2, 2, a.level, S());
    d.level = a.level, d.scale = b.scale;
}
void HEVM::op_addcp(int dst, int lhs, int rhs)
{
    hevm_ctxt &a = reg(lhs);
    hevm_ctxt &d = reg(dst);
    const Plain &p = plains.at(rhs);
    if (debug) std::cout << std::log2(a.scale) << std::log2(p.scale) << std::endl;
    a.scale = p.scale; // SEAL_HEVM.cpp:308
    if (a.level != p.level) {
        fprintf(stderr, "[dacapo_amd] addcp: level mismatch %d vs %d\n", a.level, p.level);
        abort();
    }
    launch_add_plain(*ctx, view(d), view(a), p.d, a.level, S());
    d.level = a.level, d.scale = p.scale;
}
void HEVM::op_mulcc(int dst, int lhs, int rhs)
{
    hevm_ctxt &a = reg(lhs);
    hevm_ctxt &b = reg(rhs);
    hevm_ctxt &d = reg(dst);
    if (debug) std::cout << std::log2(a.scale) << std::log2(b.scale) << std::endl;
    if (a.level != b.level) {
        fprintf(stderr, "[dacapo_amd] mulcc: level mismatch %d vs %d\n", a.level, b.level);
        abort();
    }
    mul_relin(*ctx, W(), view(d), view(a), view(b), keys.relin, a.level, S());
    n_keyswitch++, n_ntt += ks_ntts(a.level);
    const double sc = a.scale * b.scale;
    d.level = a.level, d.scale = sc;
}
void HEVM::op_mulcp(int dst, int lhs, int rhs)
{
    hevm_ctxt &a = reg(lhs);
    hevm_ctxt &d = reg(dst);
    const Plain &p = plains.at(rhs);
    if (debug) std::cout << std::log2(a.scale) << std::log2(p.scale) << std::endl;
    if (a.level != p.level) {
        fprintf(stderr, "[dacapo_amd] mulcp: level mismatch %d vs %d\n", a.level, p.level);
        abort();
    }
    launch_ew(*ctx, EwOp::Mul, view(d), view(a), CtView{ p.d, 0 }, 2, 1, a.level, S());
    const double sc = a.scale * p.scale;
    d.level = a.level, d.scale = sc;
}
// ---- extension opcodes 17-19 (hevm_asm.OP_CONJ / OP_MODRAISE / OP_SETSCALE: the pieces real CKKS bootstrapping needs, ckks_boot.py) ----
void HEVM::op_conj(int dst, int src)
{ // complex conjugation of the slots = the Galois automorphism X -> X^(2N-1), a key of the default set
    hevm_ctxt &s = reg(src);
    hevm_ctxt &d = reg(dst);
    const u32 elt = (u32)(2 * ctx->N - 1);
    if (!keys.galois.count(elt)) {
        fprintf(stderr, "[dacapo_amd] conj: no Galois key for the conjugation\n");
        abort();
    }
    rotate_hop(*ctx, W(), view(d), view(s), elt, keys.galois.at(elt), s.level, S());
    n_keyswitch++, n_ntt += ks_ntts(s.level);
    d.level = s.level, d.scale = s.scale;
}
void HEVM::op_modraise(int dst, int src, int target)
{
    hevm_ctxt &s = reg(src);
    hevm_ctxt &d = reg(dst);
    if (s.level != 1 || target < 1 || target > ctx->max_level()) {
        fprintf(stderr, "[dacapo_amd] modraise: the operand must sit at 1 prime (has %d) and the target within 1..%d\n", s.level, ctx->max_level());
        abort();
    }
    const EwItem it{ view(d), view(s), view(s) };
    modraise(*ctx, W().ks_digits, &it, 1, target, S());
    d.level = target, d.scale = s.scale;
}
void HEVM::op_setscale(int dst, int src, int const_idx)
{
    hevm_ctxt &s = reg(src);
    hevm_ctxt &d = reg(dst);
    if (dst != src) launch_ew(*ctx, EwOp::Copy, view(d), view(s), view(s), 2, 2, s.level, S());
    d.level = s.level, d.scale = buffer[(size_t)const_idx][0]; // validated by load_program
}

// host-side constants of the device CRT for level ell (built once per level)
const HEVM::CrtTables &HEVM::crt_tables(int ell)
{
    auto it = crt_.find(ell);
    if (it != crt_.end()) return it->second;
    const Context &c = *ctx;
    if (ell > kMaxCrt || ell > 16) {
        fprintf(stderr, "[dacapo_amd] device CRT supports up to 16 primes (Q must fit a double's range)\n");
        abort();
    }
    std::vector<u64> inv(ell), mmod((size_t)ell * ell, 0), hmod(ell), hdig(ell);
    std::vector<double> mdbl(ell);
    long double prod = 1.0L;
    for (int k = 0; k < ell; k++) {
        const u64 qk = c.primes[k];
        u64 acc = 1;
        for (int i = 0; i <= k; i++) {
            mmod[(size_t)i * ell + k] = acc;
            if (i < k) acc = h_mulmod(acc, c.primes[i] % qk, qk);
        }
        inv[k] = h_invmod(acc, qk);
        mdbl[k] = (double)prod;
        prod *= (long double)qk;
    }
    // h = floor(Q/2) as a multiword integer, then its residues and mixed-radix digits
    std::vector<u64> Q(ell + 1, 0);
    Q[0] = 1;
    for (int k = 0; k < ell; k++) {
        u128 carry = 0;
        for (int w = 0; w <= ell; w++) {
            u128 t = (u128)Q[w] * c.primes[k] + (u64)carry;
            Q[w] = (u64)t;
            carry = t >> 64;
        }
    }
    std::vector<u64> h(ell + 1);
    for (int w = 0; w <= ell; w++) h[w] = (Q[w] >> 1) | (w < ell ? (Q[w + 1] << 63) : 0);
    for (int k = 0; k < ell; k++) {
        u128 r = 0;
        for (int w = ell; w >= 0; w--) r = ((r << 64) | h[w]) % c.primes[k];
        hmod[k] = (u64)r;
    }
    std::vector<u64> t = h;
    for (int k = 0; k < ell; k++) { // digit k = t mod q_k ; t /= q_k
        u128 r = 0;
        for (int w = ell; w >= 0; w--) {
            u128 cur = (r << 64) | t[w];
            t[w] = (u64)(cur / c.primes[k]);
            r = cur % c.primes[k];
        }
        hdig[k] = (u64)r;
    }
    CrtTables tb;
    auto up = [&](const void *src, size_t bytes) {
        void *d = nullptr;
        DC_HIP_CHECK(vm_malloc(&d, bytes));
        DC_HIP_CHECK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        return d;
    };
    tb.inv = (u64 *)up(inv.data(), inv.size() * 8);
    tb.mmod = (u64 *)up(mmod.data(), mmod.size() * 8);
    tb.hmod = (u64 *)up(hmod.data(), hmod.size() * 8);
    tb.hdig = (u64 *)up(hdig.data(), hdig.size() * 8);
    tb.mdbl = (double *)up(mdbl.data(), mdbl.size() * 8);
    return crt_.emplace(ell, tb).first->second;
}

void HEVM::op_bootstrap(int dst, int src, int target_level)
{
    const auto t0 = std::chrono::steady_clock::now();
    hevm_ctxt &s = reg(src);
    if (debug) std::cout << std::log2(s.scale) << std::endl;
    boot_item(view(s), s.level, s.scale, reg(dst), target_level);
    t_bootstrap += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// the SEAL VM's stand-in: decrypt -> decode -> re-encode at `target_level` primes -> encrypt (SEAL_HEVM.cpp:328-333),
// evaluated on the device (see reencode_kernel).  dst may alias src.
void HEVM::boot_item(CtView src, int ell, double src_scale, hevm_ctxt &dst, int target_level)
{
    Context &c = *ctx;
    const size_t N = c.N;
    if (!keys.sk || !keys.pk || ell < 1 || target_level < 1 || target_level > c.max_level()) {
        fprintf(stderr, "[dacapo_amd] bootstrap: needs a full VM (secret + public key) and a valid target level\n");
        abort();
    }
    const double new_scale = pow(2.0, (double)(int64_t)std::log2(src_scale)); // SEAL_HEVM.cpp:332 -> :262
    const CrtTables &tb = crt_tables(ell);
    u64 *pt = W().ks_tmp;      // [ell][N]
    u64 *lohi = W().ks_digits; // [2][N]
    f_irows_decrypt(c, src, keys.sk, ell, pt, S()); // c0 + c1*s fused into the first inverse phase
    launch_ntt_cols_inv(c, pt, (long)N, ell, nullptr, 0, 0, S());
    const CrtDev cd{ tb.inv, tb.mmod, tb.hmod, tb.hdig, tb.mdbl };
    DC_LAUNCH(reencode_kernel, dim3((unsigned)((N / 2 + kVmThreads) / kVmThreads)), dim3(kVmThreads), 0, S(), lohi,
                       lohi + N, pt, ell, N, c.d_mods, cd, new_scale / src_scale);
    Plain ptx{ lanes[cur].boot_plain.d, target_level, new_scale };
    DC_LAUNCH(lift_i128_kernel, dim3((unsigned)(N / kVmThreads), (unsigned)target_level), dim3(kVmThreads), 0, S(),
                       ptx.d, lohi, lohi + N, N, c.d_mods);
    launch_ntt(c, false, ptx.d, (long)N, target_level, nullptr, 0, 0, S());
    encrypt_plain(dst, ptx);
}

// batched opcode 10, randomness half: zero-encryptions of items [first, first+B) of the plan's boot table (all with target
// level t) into their zenc slots.  7 launches whatever B is.
void HEVM::plan_zero_encrypt(int first, int B, int t, hipStream_t s)
{
    Context &c = *ctx;
    const size_t N = c.N;
    const int cnt = t + 1;
    if (!keys.pk) {
        fprintf(stderr, "[dacapo_amd] bootstrap: this VM has no public key\n");
        abort();
    }
    Plan &P = plan;
    DC_LAUNCH(sample_enc_batch_kernel, dim3((unsigned)(N / (kRngCoefs * kVmThreads)), (unsigned)(3 * B)), dim3(kVmThreads), 0, s,
                       P.boot_ue, N, cnt, rng.secret, ((u64)1 << 32) + (u64)first, c.d_mods, d_epoch);
    launch_ntt(c, false, P.boot_ue, (long)N, 3 * cnt * B, nullptr, 0, cnt, s);
    DC_LAUNCH(pk_encrypt_batch_kernel, dim3((unsigned)(N / (2 * kVmThreads)), (unsigned)cnt, (unsigned)(2 * B)), dim3(kVmThreads),
                       0, s, P.boot_tmp, keys.pk, (long)c.K * (long)N, P.boot_ue, cnt, N, c.d_mods);
    b_rescale(c, P.ws[0], P.d_boot_rs + first, B, cnt, s); // divide-and-round by the extra prime, straight into the zenc slots
}

// batched opcode 10, data half: B items at `ell` primes -> `t` primes.  5 launches.
void HEVM::plan_boot_step(int first, int B, int ell, int t, int lane, hipStream_t s, const Handoff &h)
{
    Context &c = *ctx;
    const size_t N = c.N;
    if (!keys.sk) {
        fprintf(stderr, "[dacapo_amd] bootstrap: needs a full VM (secret + public key)\n");
        abort();
    }
    Plan &P = plan;
    const BootItem *items = P.d_boot + first;
    const CrtTables &tb = crt_tables(ell);
    const CrtDev cd{ tb.inv, tb.mmod, tb.hmod, tb.hdig, tb.mdbl };
    u64 *pt = P.boot_pt[lane], *ptx = P.boot_ptx[lane];
    if (h.in) // the producer's last kernel already decrypted into the first inverse phase (plan.hpp Handoff)
        pt = const_cast<u64 *>(h.in);
    else
        f_irows_decrypt_items(c, items, P.d_sum_srcs, keys.sk, ell, pt, B, s);
    launch_ntt_cols_inv(c, pt, (long)N, B * ell, nullptr, 0, ell, s);
    if (ell == 1) // trivial composition: re-encode inside the first forward phase's loader (4 launches per batch)
        f_boot_reencode_fcols(c, pt, ptx, items, B, ell, t, cd, s);
    else {
        DC_LAUNCH(reencode_lift_batch_kernel, dim3((unsigned)((N / 2 + kVmThreads) / kVmThreads), (unsigned)B), dim3(kVmThreads), 0,
                           s, ptx, pt, items, ell, t, N, c.d_mods, cd);
        launch_ntt_cols_fwd(c, ptx, (long)N, B * t, nullptr, 0, t, s);
    }
    f_frows_boot_final(c, ptx, items, B, t, s);
}

void HEVM::dispatch(const WireOp &op)
{
    switch (op.opcode) {
    case 1: op_rotate(op.dst, op.lhs, (int16_t)op.rhs); break;
    case 2: op_negate(op.dst, op.lhs); break;
    case 3: op_rescale(op.dst, op.lhs); break;
    case 4: op_modswitch(op.dst, op.lhs, (int16_t)op.rhs); break;
    case 5: fprintf(stderr, "This VM does not support native upscale op\n"); abort();
    case 6: op_addcc(op.dst, op.lhs, op.rhs); break;
    case 7: op_addcp(op.dst, op.lhs, op.rhs); break;
    case 8: op_mulcc(op.dst, op.lhs, op.rhs); break;
    case 9: op_mulcp(op.dst, op.lhs, op.rhs); break;
    case 10: op_bootstrap(op.dst, op.lhs, op.rhs); break;
    case kOpConj: op_conj(op.dst, op.lhs); break;
    case kOpModRaise: op_modraise(op.dst, op.lhs, op.rhs); break;
    case kOpSetScale: op_setscale(op.dst, op.lhs, op.rhs); break;
    default: break; // 0 = Encode (done in preprocess), 0xFFFF buffer marker and unknown opcodes are no-ops
    }
}

// The dispatch loop of SEAL_HEVM::run (SEAL_HEVM.cpp:336-401): one instruction at a time, in program order, on one stream.
void HEVM::execute()
{
    memset(op_counts, 0, sizeof(op_counts));
    n_keyswitch = n_ntt = 0;
    t_bootstrap = 0.0;
    int i = (int)((header.hevm_header_size + config.config_body_length) / 8), j = 0;
    for (const WireOp &op : ops) {
        if (debug) {
            std::cout << std::endl;
            std::cout << std::oct << i++ << " " << std::dec << j++ << std::endl;
            std::cout << "opcode [" << op.opcode << "], dst [" << op.dst << "], lhs [" << op.lhs << "], rhs [" << op.rhs << "]"
                      << std::endl;
        }
        if (op.opcode <= 10) op_counts[op.opcode]++;
        if (op.opcode == 0 || (op.opcode > 10 && (op.opcode < kOpConj || op.opcode > kOpSetScale))) continue;
        dispatch(op);
    }
    bump_epoch(S());
}

void HEVM::run()
{
    if (use_plan && !debug) {
        run_plan();
        return;
    }
    if (streams != 1) {
        fprintf(stderr, "[dacapo_amd] several ciphertext streams need the batched plan (option plan = 1, no debug)\n");
        abort();
    }
    execute();
    DC_HIP_CHECK(hipStreamSynchronize(S())); // the caller's timer stops when run() returns
}

} // namespace dacapo

// =========================================================================================================
// the 18 symbols of the reference (SEAL_HEVM.cpp:404-504) + extensions
// =========================================================================================================
namespace dacapo {
thread_local VmAllocs *g_vm_allocs = nullptr;

void HEVM::destroy_device_state()
{
    (void)hipDeviceSynchronize();
    drop_plan_graph();
    for (Lane &ln : lanes)
        if (ln.stream) (void)hipStreamDestroy(ln.stream), ln.stream = nullptr;
    if (aux_stream) (void)hipStreamDestroy(aux_stream), aux_stream = nullptr;
    for (hipEvent_t e : plan.events) (void)hipEventDestroy(e);
    plan.events.clear();
    const std::vector<void *> live(allocs.live.begin(), allocs.live.end());
    allocs.live.clear();
    for (void *p : live) (void)hipFree(p);
}
} // namespace dacapo

using dacapo::HEVM;
// every entry point names the VM whose allocations it may create or release
static HEVM *V(void *vm)
{
    HEVM *h = static_cast<HEVM *>(vm);
    dacapo::g_vm_allocs = &h->allocs;
    return h;
}

static std::vector<char> slurp(const char *path)
{
    std::ifstream f(path, std::ios::in | std::ios::binary | std::ios::ate);
    if (!f) {
        fprintf(stderr, "[dacapo_amd] cannot open %s\n", path);
        abort();
    }
    std::vector<char> buf((size_t)f.tellg());
    f.seekg(0);
    f.read(buf.data(), (std::streamsize)buf.size());
    return buf;
}

static void env_params(int &logN, int &K)
{ // the reference hard-codes N = 15, L = 14 (SEAL_HEVM.cpp:39-40) -- the defaults of options logn / primes; tests shrink the ring
    logN = (int)dacapo::option(dacapo::OPT_LOGN), K = (int)dacapo::option(dacapo::OPT_PRIMES);
}

extern "C" {

void *initFullVM(char *dir, bool device)
{
    (void)device;
    auto vm = new HEVM();
    vm->load_keys(dir, true, true, true);
    return vm;
}
void *initClientVM(char *dir)
{
    auto vm = new HEVM();
    vm->load_keys(dir, true, true, false);
    return vm;
}
void *initServerVM(char *dir)
{
    auto vm = new HEVM();
    vm->load_keys(dir, false, false, true);
    return vm;
}
void create_context(char *dir)
{
    int logN, K;
    env_params(logN, K);
    HEVM vm;
    vm.init_context(logN, K, nullptr);
    vm.generate_keys(dacapo::rng_keys_from_os(), true, true, true);
    vm.save_keys(dir);
    vm.destroy_device_state(); // a temporary VM: its keys live in the files now
    dacapo::g_vm_allocs = nullptr;
}
void load(void *vm, char *constant, char *vmfile)
{
    auto hevm = V(vm);
    std::vector<char> cst = slurp(constant), prog = slurp(vmfile);
    hevm->load_constants(cst.data(), cst.size());
    hevm->load_program(prog.data(), prog.size(), false);
}
void loadClient(void *vm, void *is)
{
    auto hevm = V(vm);
    std::vector<char> prog = slurp(static_cast<const char *>(is));
    hevm->load_program(prog.data(), prog.size(), true);
    hevm->reset_res_dst();
}
void encrypt(void *vm, int64_t i, double *dat, int len) { V(vm)->encrypt(i, dat, len); }
void decrypt(void *vm, int64_t i, double *dat) { V(vm)->decrypt(i, dat); }
void decrypt_result(void *vm, int64_t i, double *dat)
{
    auto hevm = V(vm);
    hevm->decrypt((int64_t)hevm->res_dst.at((size_t)i), dat);
}
int64_t getResIdx(void *vm, int64_t i) { return (int64_t) V(vm)->res_dst.at((size_t)i); }
void *getCtxt(void *vm, int64_t id) { return &V(vm)->reg((size_t)id); }
void preprocess(void *vm) { V(vm)->preprocess(); }
void run(void *vm) { V(vm)->run(); }
int64_t getArgLen(void *vm) { return (int64_t) V(vm)->header.arg_length; }
int64_t getResLen(void *vm) { return (int64_t) V(vm)->header.res_length; }
void setDebug(void *vm, bool enable) { V(vm)->debug = enable; }
void setToGPU(void *vm, bool ongpu)
{
    (void)vm;
    (void)ongpu; // always resident on the MI355X
}
void printMem(void *vm) { (void)vm; } // O(1): it sits inside the caller's timed region (runner.py:223-225)

// ---- extensions -----------------------------------------------------------------------------------------
void *hevm_init_fresh(int logN, int num_primes)
{ // create_context + initFullVM without the files: parameters and the reference's key set generated in HBM from the OS's randomness
    int dl, dk;
    env_params(dl, dk);
    auto vm = new HEVM();
    vm->init_context(logN > 0 ? logN : dl, num_primes > 0 ? num_primes : dk, nullptr);
    vm->generate_keys(dacapo::rng_keys_from_os(), true, true, true);
    return vm;
}
void *hevm_init_fresh_primes(int logN, const uint64_t *primes, int num_primes)
{ // the same on an explicit chain (each prime = 1 mod 2N, 45..60 bits; other than 60: the generic-width build), e.g. a HEaaN-style mixed one
    auto vm = new HEVM();
    vm->init_context(logN, num_primes, primes);
    vm->generate_keys(dacapo::rng_keys_from_os(), true, true, true);
    return vm;
}
void *hevm_context(void *vm)
{ // borrowed kernel-level handle, one per VM (never freed, like the VM itself)
    auto h = V(vm);
    if (!h->ckks_handle) h->ckks_handle = new dc_context{ h->ctx.get(), false };
    return h->ckks_handle;
}
const uint64_t *hevm_relin_key(void *vm) { return V(vm)->keys.relin; }
const uint64_t *hevm_galois_key(void *vm, uint32_t elt)
{
    auto &g = V(vm)->keys.galois;
    auto it = g.find(elt);
    return it == g.end() ? nullptr : it->second;
}
const uint64_t *hevm_public_key(void *vm) { return V(vm)->keys.pk; }
const uint64_t *hevm_plain(void *vm, int64_t i, int32_t *level, double *scale)
{
    const dacapo::Plain &p = V(vm)->plains.at((size_t)i);
    if (level) *level = p.level;
    if (scale) *scale = p.scale;
    return p.d;
}
const uint64_t *hevm_plain_special(void *vm, int64_t i) { return V(vm)->plains.at((size_t)i).dsp; }
void hevm_load_mem(void *vm, const void *cst, uint64_t cst_len, const void *hevm, uint64_t hevm_len)
{
    auto h = V(vm);
    h->load_constants(cst, cst_len);
    h->load_program(hevm, hevm_len, false);
}
void hevm_set_streams(void *vm, int n) { V(vm)->set_streams(n); }
void hevm_select_stream(void *vm, int s) { V(vm)->select_stream(s); }
double hevm_last_run_bootstrap_seconds(void *vm) { return V(vm)->t_bootstrap; }
uint64_t hevm_plaintext_bytes(void *vm)
{ // HBM held for the program's plaintexts: the pre-encoded pool, or (on-line encode) the resident constants + the encode window
    auto h = V(vm);
    if (h->online_encode) return h->online.const_bytes + h->plan.enc_arena_bytes + h->plan.enc_scratch_bytes;
    uint64_t total = 0;
    for (const dacapo::Plain &p : h->plains) total += (uint64_t)p.level * h->ctx->N * 8;
    return total;
}
void hevm_add_rotation_keys(void *vm, const int64_t *offsets, int count)
{ // KeyGenerator::create_galois_keys(steps): a direct key per slot offset (left = positive), next to the default +-2^k set
    auto h = V(vm);
    if (!h->keys.sk) {
        fprintf(stderr, "[dacapo_amd] hevm_add_rotation_keys: this VM holds no secret key\n");
        abort();
    }
    const int64_t slots = (int64_t)(h->ctx->N >> 1);
    for (int i = 0; i < count; i++) {
        int64_t st = offsets[i] % slots; // the HEaaN runtime lists left rotations in [1, slots) (HEAAN_HEVM.cpp:58-64)
        if (st > slots / 2) st -= slots;
        if (st < -slots / 2) st += slots;
        if (st == 0) continue;
        const uint32_t elt = dc_galois_elt_from_step(static_cast<dc_context *>(hevm_context(vm)), (int)st);
        h->add_galois_key(elt);
    }
    DC_HIP_CHECK(hipStreamSynchronize(h->S()));
    h->plan.ready = false; // rotations by these offsets are single hops from now on: the plan of a loaded program is rebuilt
    h->drop_plan_graph();
}
// ---- key replication across VM replicas (evaluation keys are read-only and replicated over the GPUs) ------------------
using namespace dacapo;
// wrapping checksum of `words` 64-bit words: sum of x[i] * (2 i + 1)
__global__ __launch_bounds__(256) void key_checksum_kernel(const u64 *__restrict__ x, size_t words, unsigned long long *__restrict__ out)
{
    __shared__ u64 part[256];
    u64 acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) acc += x[i] * (2 * (u64)i + 1);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)part[0]);
}
// Enumerates the VM's key buffers in a canonical order (secret, public, relinearisation, Galois keys by ascending element): fills
// ptrs[i] / words[i] for up to `cap` buffers and returns how many there are.  Replicas that hold the same key set list the same sizes.
int hevm_key_buffers(void *vm, uint64_t **ptrs, uint64_t *words, int cap)
{
    auto h = V(vm);
    const Context &c = *h->ctx;
    std::vector<std::pair<u64 *, size_t>> v;
    if (h->keys.sk) v.push_back({ h->keys.sk, (size_t)c.K * c.N });
    if (h->keys.pk) v.push_back({ h->keys.pk, (size_t)2 * c.K * c.N });
    if (h->keys.relin) v.push_back({ h->keys.relin, h->key_elems() });
    std::vector<u32> elts;
    for (const auto &kv : h->keys.galois) elts.push_back(kv.first);
    std::sort(elts.begin(), elts.end());
    for (u32 e : elts) v.push_back({ h->keys.galois.at(e), h->key_elems() });
    for (int i = 0; i < (int)v.size() && i < cap; i++) ptrs[i] = v[(size_t)i].first, words[i] = (uint64_t)v[(size_t)i].second;
    return (int)v.size();
}
// one 64-bit digest of all key material (order as above), computed on the device
uint64_t hevm_key_digest(void *vm)
{
    auto h = V(vm);
    const int n = hevm_key_buffers(vm, nullptr, nullptr, 0);
    std::vector<uint64_t *> ptrs((size_t)n);
    std::vector<uint64_t> words((size_t)n);
    hevm_key_buffers(vm, ptrs.data(), words.data(), n);
    unsigned long long *d = nullptr, out = 0;
    DC_HIP_CHECK(hipMalloc(&d, 8));
    u64 digest = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < n; i++) {
        DC_HIP_CHECK(hipMemsetAsync(d, 0, 8, h->S()));
        DC_LAUNCH(key_checksum_kernel, dim3(1024), dim3(256), 0, h->S(), ptrs[(size_t)i], (size_t)words[(size_t)i], d);
        DC_HIP_CHECK(hipMemcpyAsync(&out, d, 8, hipMemcpyDeviceToHost, h->S()));
        DC_HIP_CHECK(hipStreamSynchronize(h->S()));
        digest = (digest ^ (u64)out) * 0xBF58476D1CE4E5B9ull + (u64)words[(size_t)i];
    }
    (void)hipFree(d);
    return digest;
}
// after key buffers have been overwritten from outside (a broadcast from another replica): forget everything derived from them
void hevm_keys_replaced(void *vm)
{
    auto h = V(vm);
    DC_HIP_CHECK(hipStreamSynchronize(h->S()));
    h->plan.ready = false;
    h->drop_plan_graph();
}
void hevm_save_ctxt(void *vm, int64_t reg, const char *path) { V(vm)->save_ctxt((size_t)reg, path); }
void hevm_load_ctxt(void *vm, int64_t reg, const char *path) { V(vm)->load_ctxt((size_t)reg, path); }

// ---- host-only entry points (no GPU involved) ---------------------------------------------------------------
void hevm_seal_parms_id(uint64_t poly_modulus_degree, const uint64_t *primes, int count, uint64_t out[4])
{
    const dacapo::sealio::ParmsId id = dacapo::sealio::parms_id(poly_modulus_degree, primes, (size_t)count);
    memcpy(out, id.data(), 32);
}
void hevm_seal_save_parms(const char *path, int compr_mode, uint64_t poly_modulus_degree, const uint64_t *primes, int count)
{
    namespace sio = dacapo::sealio;
    sio::Writer w;
    sio::Params p;
    p.N = poly_modulus_degree, p.primes.assign(primes, primes + count);
    put_params(w, p);
    write_object_file(path, w.buf, (sio::Compr)compr_mode);
}
int hevm_seal_load_parms(const char *path, uint64_t *poly_modulus_degree, uint64_t *primes, int capacity)
{
    namespace sio = dacapo::sealio;
    const std::vector<uint8_t> file = sio::read_file(path);
    std::vector<uint8_t> owned;
    sio::Reader outer(file.data(), file.size(), path);
    sio::Reader m = open_object(outer, owned);
    const sio::Params p = get_params(m);
    if (p.scheme != sio::kSchemeCkks) m.fail("scheme is not CKKS");
    *poly_modulus_degree = p.N;
    for (size_t i = 0; i < p.primes.size() && (int)i < capacity; i++) primes[i] = p.primes[i];
    return (int)p.primes.size();
}
void hevm_seal_save_ciphertext(const char *path, int compr_mode, uint64_t poly_modulus_degree, const uint64_t *primes, int limbs, int size,
                               int is_ntt, double scale, const uint64_t *data)
{
    namespace sio = dacapo::sealio;
    sio::Writer w;
    sio::CtHeader h;
    h.id = sio::parms_id(poly_modulus_degree, primes, (size_t)limbs);
    h.is_ntt = is_ntt != 0, h.size = (uint64_t)size, h.N = poly_modulus_degree, h.limbs = (uint64_t)limbs, h.scale = scale;
    put_ciphertext(w, h, data);
    write_object_file(path, w.buf, (sio::Compr)compr_mode);
}
int64_t hevm_seal_load_ciphertext(const char *path, uint64_t *poly_modulus_degree, int *limbs, int *size, int *is_ntt, double *scale,
                                  uint64_t parms_id[4], uint64_t *data, uint64_t capacity)
{
    namespace sio = dacapo::sealio;
    const std::vector<uint8_t> file = sio::read_file(path);
    std::vector<uint8_t> owned;
    sio::Reader outer(file.data(), file.size(), path);
    sio::Reader m = open_object(outer, owned);
    const uint64_t *src = nullptr;
    const sio::CtHeader h = get_ciphertext(m, src);
    *poly_modulus_degree = h.N, *limbs = (int)h.limbs, *size = (int)h.size, *is_ntt = h.is_ntt ? 1 : 0, *scale = h.scale;
    memcpy(parms_id, h.id.data(), 32);
    const uint64_t words = h.size * h.N * h.limbs;
    if (data && words <= capacity) memcpy(data, src, (size_t)words * 8);
    return (int64_t)words;
}
int hevm_seal_zstd_available(void) { return dacapo::sealio::zstd_available() ? 1 : 0; }
void hevm_chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t nonce, uint32_t out[16])
{
    dacapo::ChaChaKey k;
    memcpy(k.w, key, 32);
    uint32_t o[16];
    dacapo::chacha20_block(k, counter, nonce, o);
    memcpy(out, o, 64);
}
void hevm_chacha20_blocks_device(const uint32_t key[8], uint64_t counter, uint64_t nonce, int blocks, uint32_t *out_host)
{ // the same block function as the samplers run it, on the GPU (parity test of the device code path)
    dacapo::ChaChaKey k;
    memcpy(k.w, key, 32);
    uint32_t *d = nullptr;
    DC_HIP_CHECK(hipMalloc(&d, (size_t)blocks * 64));
    DC_LAUNCH(dacapo::chacha_blocks_kernel, dim3((unsigned)((blocks + 63) / 64)), dim3(64), 0, 0, d, k, counter, nonce, blocks);
    DC_HIP_CHECK(hipMemcpy(out_host, d, (size_t)blocks * 64, hipMemcpyDeviceToHost));
    (void)hipFree(d);
}

void hevm_last_run_stats(void *vm, int64_t *op_counts, int64_t *keyswitches, int64_t *ntts)
{
    auto h = V(vm);
    if (op_counts) memcpy(op_counts, h->op_counts, sizeof(h->op_counts));
    if (keyswitches) *keyswitches = h->n_keyswitch;
    if (ntts) *ntts = h->n_ntt;
}

// option hyb_lazy_sum: which rotate instructions of the loaded program the plan of the last run() executed as lazy sums (one division by P
// per group).  out = [n_0, op, ..., op, n_1, op, ...]: per group its size and the instruction indices of its rotations, ascending.  Returns
// the number of int32 values the list takes (written only if cap is large enough), 0 without groups, -1 before the first run().
int64_t hevm_plan_lazy_groups(void *vm, int32_t *out, int64_t cap)
{
    auto h = V(vm);
    if (!h->plan.ready) return -1;
    std::vector<int32_t> flat;
    for (const auto &p : h->plan.pops)
        if (!p.dead && p.kind == HEVM::P_ROTSUM) {
            flat.push_back((int32_t)p.ops.size());
            flat.insert(flat.end(), p.ops.begin(), p.ops.end());
        }
    if (out && cap >= (int64_t)flat.size()) memcpy(out, flat.data(), flat.size() * sizeof(int32_t));
    return (int64_t)flat.size();
}

// the reference never frees a VM (no destroy symbol); a long-lived host that creates many can return one's HBM with this
void hevm_destroy(void *vm)
{
    if (!vm) return;
    HEVM *h = V(vm);
    h->destroy_device_state();
    dacapo::g_vm_allocs = nullptr;
    delete static_cast<dc_context *>(h->ckks_handle);
    delete h;
}

} // extern "C"
