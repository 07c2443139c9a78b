// Single-crossing negacyclic NTT for the reference ring (N = 2^15): one workgroup of 1024 threads owns a whole 256 KiB limb --
// 32 coefficients per thread in VGPRs (half of the CU's 512 KB register file), read once from HBM, written once.  The two-launch
// tiles of ntt_tile.hpp move every limb through HBM twice (profiles/r02_ntt_hbm_traffic.json: 2.0x the algorithmic bytes); for
// launches with thousands of limbs (convolution-layer batches, bootstrapping, the roofline leg) that second crossing is the
// largest avoidable cost.  Same butterflies (ntt_tile.hpp: ct_bfly / gs_bfly, fold schedule by global stage), same tables, every
// output canonical: results are bit-identical to the two-launch transform (tests/test_gpu_ntt.py).
//
// Index algebra.  A coefficient index is (a, b, c) = bits 14..10, 9..5, 4..0.  Three radix-32 passes, each on one 5-bit field held
// in a thread's 32 registers while the other ten bits are the thread:
//     forward   pass A: stages 0..4   regs = a   thread = (b, c)   twiddles tw[2^s + (a >> ..)]: wave-uniform -> scalar loads
//               -- exchange 1 (through LDS, across waves): regs a <-> thread bits b
//               pass B: stages 5..9   regs = b   thread = (a, c)   twiddles depend on a: two addresses per wave instruction
//               -- exchange 2 (inside the wavefront, lane_swap: v_permlane16_swap / DPP): regs b <-> lane bits c
//               pass C: stages 10..14 regs = c   thread = (a, b)   twiddles per thread, contiguous over the lanes
//               -- exchange 3 (inside the wavefront): regs c <-> lane bits b, so that the store is lane-contiguous
//     inverse   the mirror image (Gentleman-Sande stages 14..0, N^-1 merged into the last one).
// HBM accesses are 8 bytes per lane, 256- or 512-byte contiguous segments per wave instruction.
//
// Exchange 1 moves the whole limb (256 KiB) through the 160 KiB LDS in two rounds: registers whose destination is one of the first
// ten waves (20 of the 32 register indices: exactly 160 KiB) first, the other twelve after those waves have read.  A thread then
// holds at most 12 old + 32 new coefficients (88 VGPRs); the kernel is built for 128 VGPRs = 16 waves per CU = one workgroup.
#include "kernels.hpp"
#include "lane_xchg.hpp"
#include "ntt_tile.hpp"

namespace dacapo {

constexpr int kFullLogN = 15;
constexpr int kFullThreads = 1024;
constexpr int kFullLdsElems = 20 * 1024; // 160 KiB

// ---- pass A (stages 0..4): the twiddle of a butterfly depends on the register index only -> scalar loads ---------------------------------
template <bool INV, int U>
__device__ __forceinline__ void full_stage_a(u64 (&x)[32], const u64 *__restrict__ tw, const DModulus &M)
{
    if constexpr (INV && U == 0) { // the very last inverse stage carries N^-1
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const u64 sv = x[e] + x[e | 16], d = x[e] + (M.q << 2) - x[e | 16];
            x[e] = mulmod_lazy(M.inv_n, sv, M.delta);
            x[e | 16] = mulmod_lazy(M.inv_n_w, d, M.delta);
        }
    } else {
#pragma unroll
        for (int g = 0; g < (1 << U); g++) {
            const u64 w = tw[(1u << U) + (u32)g];
#if defined(DC_FULL_NO_BFLY) // timing experiment only (wrong results): everything but the arithmetic
            x[g << (5 - U)] += w;
#else
#pragma unroll
            for (int e = 0; e < (16 >> U); e++) {
                const int j0 = (g << (5 - U)) | e, j1 = j0 | (16 >> U);
                if constexpr (!INV)
                    ct_bfly(x[j0], x[j1], w, M, fwd_stage_folds(U));
                else
                    gs_bfly(x[j0], x[j1], w, M);
            }
#endif
        }
    }
}
template <bool INV>
__device__ __forceinline__ void full_pass_a(u64 (&x)[32], const u64 *__restrict__ tw, const DModulus &M)
{
    if constexpr (!INV) {
        full_stage_a<false, 0>(x, tw, M);
        full_stage_a<false, 1>(x, tw, M);
        full_stage_a<false, 2>(x, tw, M);
        full_stage_a<false, 3>(x, tw, M);
        full_stage_a<false, 4>(x, tw, M);
    } else {
        full_stage_a<true, 4>(x, tw, M);
        full_stage_a<true, 3>(x, tw, M);
        full_stage_a<true, 2>(x, tw, M);
        full_stage_a<true, 1>(x, tw, M);
        full_stage_a<true, 0>(x, tw, M);
    }
}

// ---- passes B and C (stages S0..S0+4, S0 = 5 / 10): the twiddles depend on the thread (hi = the index bits above the pass's field).
// A stage's twiddles are requested one or two stages before its butterflies (the compiler otherwise issues each load ~80 instructions
// before its use, a fraction of the L2 latency, and all sixteen waves of the workgroup wait together).  The widest stage (16
// twiddles) loads only the even-indexed ones: in the bit-reversed table tw[2k + 1] = tw[2k] * tw[1] (tw[1] = psi^(N/2), a square
// root of -1), one multiplication by a wave-uniform constant instead of 16 more registers in flight.
template <bool INV>
__device__ __forceinline__ void full_stage(u64 (&x)[32], int s, int u, int g, u64 w, const DModulus &M)
{
#if defined(DC_FULL_NO_BFLY)
    x[g << (5 - u)] += w;
#else
    const int half = 16 >> u;
#pragma unroll
    for (int e = 0; e < half; e++) {
        const int j0 = (g << (5 - u)) | e, j1 = j0 | half;
        if (!INV)
            ct_bfly(x[j0], x[j1], w, M, fwd_stage_folds(s));
        else
            gs_bfly(x[j0], x[j1], w, M);
    }
#endif
}

// forward passes B and C with each twiddle loaded where it is used (the compiler's own schedule: measured faster than the pipelined
// form below for the forward direction, 894 vs 916 us on 4096 limbs; slower for the inverse, 1138 vs 1080 us)
template <int S0>
__device__ __forceinline__ void full_fwd_pass_simple(u64 (&x)[32], u32 hi, const u64 *__restrict__ tw, const DModulus &M)
{
#pragma unroll
    for (int u = 0; u < 5; u++) {
#pragma unroll
        for (int g = 0; g < (1 << u); g++) full_stage<false>(x, S0 + u, u, g, tw[(1u << (S0 + u)) + (hi << u) + (u32)g], M);
    }
}

struct FullTw {
    u64 w0, w1[2], w2[4], w3[8], w4[8];
};
__device__ __forceinline__ void full_ld2(const u64 *__restrict__ p, u64 &a, u64 &b)
{
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p);
    a = v.x, b = v.y;
}
template <int S0>
__device__ __forceinline__ void full_tw_small(FullTw &t, u32 hi, const u64 *__restrict__ tw) // stages u = 0, 1, 2: 7 words
{
    t.w0 = tw[(1u << S0) + hi];
    full_ld2(tw + (2u << S0) + (hi << 1), t.w1[0], t.w1[1]);
    full_ld2(tw + (4u << S0) + (hi << 2), t.w2[0], t.w2[1]);
    full_ld2(tw + (4u << S0) + (hi << 2) + 2, t.w2[2], t.w2[3]);
}
template <int S0>
__device__ __forceinline__ void full_tw_u3(FullTw &t, u32 hi, const u64 *__restrict__ tw)
{
#pragma unroll
    for (int i = 0; i < 4; i++) full_ld2(tw + (8u << S0) + (hi << 3) + 2 * i, t.w3[2 * i], t.w3[2 * i + 1]);
}
template <int S0>
__device__ __forceinline__ void full_tw_u4(FullTw &t, u32 hi, const u64 *__restrict__ tw)
{
#pragma unroll
    for (int m = 0; m < 8; m++) t.w4[m] = tw[(16u << S0) + (hi << 4) + 2 * m];
}
// keeps the loads above where the source puts them: the value is "used" here as far as the scheduler can tell
__device__ __forceinline__ void full_pin(u64 &v) { asm volatile("" : "+v"(v)); }

template <int S0, bool INV>
__device__ __forceinline__ void full_pass_bc(u64 (&x)[32], FullTw &t, u32 hi, const u64 *__restrict__ tw, u64 im, const DModulus &M)
{
    if (!INV) { // t holds u = 0, 1, 2 on entry
        full_stage<false>(x, S0, 0, 0, t.w0, M);
        full_tw_u3<S0>(t, hi, tw);
#pragma unroll
        for (int g = 0; g < 2; g++) full_stage<false>(x, S0 + 1, 1, g, t.w1[g], M);
#pragma unroll
        for (int g = 0; g < 4; g++) full_stage<false>(x, S0 + 2, 2, g, t.w2[g], M);
        full_tw_u4<S0>(t, hi, tw);
#pragma unroll
        for (int g = 0; g < 8; g++) full_stage<false>(x, S0 + 3, 3, g, t.w3[g], M);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            full_stage<false>(x, S0 + 4, 4, 2 * m, t.w4[m], M);
            full_stage<false>(x, S0 + 4, 4, 2 * m + 1, canon(mulmod_lazy(im, t.w4[m], M.delta), M), M);
        }
    } else { // t holds u = 4 (even) on entry
        full_tw_u3<S0>(t, hi, tw);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            full_stage<true>(x, S0 + 4, 4, 2 * m, t.w4[m], M);
            full_stage<true>(x, S0 + 4, 4, 2 * m + 1, canon(mulmod_lazy(im, t.w4[m], M.delta), M), M);
        }
        full_tw_small<S0>(t, hi, tw);
#pragma unroll
        for (int g = 0; g < 8; g++) full_stage<true>(x, S0 + 3, 3, g, t.w3[g], M);
#pragma unroll
        for (int g = 0; g < 4; g++) full_stage<true>(x, S0 + 2, 2, g, t.w2[g], M);
#pragma unroll
        for (int g = 0; g < 2; g++) full_stage<true>(x, S0 + 1, 1, g, t.w1[g], M);
        full_stage<true>(x, S0, 0, 0, t.w0, M);
    }
}

// 32 x 32 transpose between the register index and lane bits 0..4, inside the wavefront: register bit k <-> lane bit k
__device__ __forceinline__ void full_transpose(u64 (&x)[32])
{
#if defined(DC_FULL_NO_TRANSPOSE) // timing experiment only (wrong results)
    return;
#endif
#pragma unroll
    for (int j = 0; j < 32; j++)
        if (!(j & 16)) lane_swap<16>(x[j], x[j | 16]);
#pragma unroll
    for (int j = 0; j < 32; j++)
        if (!(j & 8)) lane_swap<8>(x[j], x[j | 8]);
#pragma unroll
    for (int j = 0; j < 32; j++)
        if (!(j & 4)) lane_swap<4>(x[j], x[j | 4]);
#pragma unroll
    for (int j = 0; j < 32; j++)
        if (!(j & 2)) lane_swap<2>(x[j], x[j | 2]);
#pragma unroll
    for (int j = 0; j < 32; j++)
        if (!(j & 1)) lane_swap<1>(x[j], x[j | 1]);
}

// Exchange 1.  Before: thread (f = wave * 2 + (lane >> 5), c = lane & 31) holds the element whose register field is r in x[r].
// After: thread (f', c) holds in y[r'] the element that thread (r', c) had in x[f'].  (forward: f = b, r = a; inverse: f = a, r = b)
__device__ __forceinline__ void full_exchange(u64 (&y)[32], const u64 (&x)[32], u64 *__restrict__ lds)
{
#if defined(DC_FULL_NO_XCHG) // timing experiment only (wrong results)
#pragma unroll
    for (int j = 0; j < 32; j++) y[j] = x[j];
    return;
#endif
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, f = (wave << 1) | (lane >> 5);
    // image "for the reader": [r' = f of the writer][reader thread] ; round 1 holds the readers of waves 0..9 (r < 20)
#pragma unroll
    for (int r = 0; r < 20; r++) lds[f * 640 + r * 32 + c] = x[r];
    __syncthreads();
    if (wave < 10) {
#pragma unroll
        for (int j = 0; j < 32; j++) y[j] = lds[j * 640 + wave * 64 + lane];
    }
    __syncthreads();
#pragma unroll
    for (int r = 20; r < 32; r++) lds[f * 384 + (r - 20) * 32 + c] = x[r];
    __syncthreads();
    if (wave >= 10) {
#pragma unroll
        for (int j = 0; j < 32; j++) y[j] = lds[j * 384 + (wave - 10) * 64 + lane];
    }
}

template <bool INV>
__global__ __launch_bounds__(kFullThreads) void ntt_full15_kernel(u64 *__restrict__ data, long limb_stride, const int *__restrict__ prime_idx,
                                                                   int prime_base, int prime_period, const DModulus *__restrict__ mods,
                                                                   const u64 *__restrict__ tw_all)
{
    // One limb per workgroup, no loop: a persistent form (grid = 256 or 512 workgroups walking over the limbs, so that one limb's stores
    // overlap the next one's loads) was measured slower, 1057-1128 us against 894 on 4096 limbs -- the loop makes the compiler keep its
    // invariants in registers and spill 71-94 VGPRs (profiles/r03_experiments.txt).
    __shared__ __attribute__((aligned(16))) u64 lds[kFullLdsElems];
    const int limb = blockIdx.x;
    const int p = prime_idx ? prime_idx[limb % prime_period] : prime_base + (limb % prime_period);
    u64 *__restrict__ d = data + (long)limb * limb_stride;
    const DModulus M = mods[p];
    const u64 *__restrict__ tw = tw_all + ((size_t)p << kFullLogN);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 31;
    const u32 f = (u32)((wave << 1) | (lane >> 5)); // the 5-bit field this thread carries in exchange 1 / passes B and C
    u64 x[32], y[32];
    FullTw t;
    const u64 im = tw[1]; // psi^(N/2) (forward table) or its inverse (inverse table): wave-uniform
    if (!INV) {
#pragma unroll
        for (int j = 0; j < 32; j++) x[j] = d[j * 1024 + tid]; // regs = a, thread = (b, c)
        full_pass_a<false>(x, tw, M);
        full_exchange(y, x, lds);                              // regs = b, thread = (a = f, c)
#if defined(DC_FULL_FWD_PIPELINED)
        full_tw_small<5>(t, f, tw);
        full_pass_bc<5, false>(y, t, f, tw, im, M);
        full_transpose(y);                                     // regs = c, lane bits 0..4 = b
        const u32 hc = (f << 5) | (u32)lo;
        full_tw_small<10>(t, hc, tw);
        full_pass_bc<10, false>(y, t, hc, tw, im, M);
#else
        full_fwd_pass_simple<5>(y, f, tw, M);
        full_transpose(y);                                     // regs = c, lane bits 0..4 = b
        full_fwd_pass_simple<10>(y, (f << 5) | (u32)lo, tw, M);
#endif
        full_transpose(y);                                     // regs = b, lane bits 0..4 = c
#pragma unroll
        for (int j = 0; j < 32; j++) d[(int)f * 1024 + j * 32 + lo] = canon(y[j], M);
    } else {
        const u32 hc = (f << 5) | (u32)lo;
        full_tw_u4<10>(t, hc, tw);
#pragma unroll
        for (int j = 0; j < 32; j++) x[j] = d[(int)f * 1024 + j * 32 + lo]; // regs = b, thread = (a = f, c)
        full_transpose(x);                                     // regs = c, lane bits 0..4 = b
        full_pass_bc<10, true>(x, t, hc, tw, im, M);
        full_tw_u4<5>(t, f, tw);
        full_transpose(x);                                     // regs = b, lane bits 0..4 = c
        full_pass_bc<5, true>(x, t, f, tw, im, M);
        full_exchange(y, x, lds);                              // regs = a, thread = (b = f, c)
        full_pass_a<true>(y, tw, M);
#pragma unroll
        for (int j = 0; j < 32; j++) d[j * 1024 + tid] = canon(y[j], M);
    }
}

bool ntt_full_supported(const Context &c) { return c.logN == kFullLogN; }
long ntt_full_min_limbs(bool inverse)
{ // Below this many limbs the two-launch tiles (16 workgroups per limb, several per CU) are faster: measured forward 148 vs 121 us at
  // 512 limbs, 250-259 vs 270-278 at 1024, 894 vs 1180 at 4096; inverse 280 vs 249-253 at 1024, 1080 vs 1069 at 4096 -- the inverse
  // only ties, so it keeps the two-launch form unless asked (profiles/r03_ntt_full.txt).  0 = never.
    static const long f = getenv("DACAPO_NTT_FULL_MIN_LIMBS") ? atol(getenv("DACAPO_NTT_FULL_MIN_LIMBS")) : 1024;
    static const long i = getenv("DACAPO_NTT_FULL_INV_MIN_LIMBS") ? atol(getenv("DACAPO_NTT_FULL_INV_MIN_LIMBS")) : 0;
    return inverse ? i : f;
}

void launch_ntt_full(const Context &c, bool inverse, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                     int prime_period, hipStream_t s)
{
    if (count <= 0) return;
    if (prime_period <= 0) prime_period = 1 << 30;
    const unsigned grid = (unsigned)count;
    if (!inverse)
        hipLaunchKernelGGL(ntt_full15_kernel<false>, dim3(grid), dim3(kFullThreads), 0, s, data, limb_stride, d_prime_idx, prime_base,
                           prime_period, c.d_mods, c.d_tw);
    else
        hipLaunchKernelGGL(ntt_full15_kernel<true>, dim3(grid), dim3(kFullThreads), 0, s, data, limb_stride, d_prime_idx, prime_base,
                           prime_period, c.d_mods, c.d_itw);
}

} // namespace dacapo
