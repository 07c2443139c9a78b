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

// stages S0 .. S0+4 (forward, Cooley-Tukey) on the 32 registers; hi = the index bits above this pass's field
template <int S0>
__device__ __forceinline__ void full_fwd_pass(u64 (&x)[32], u32 hi, const u64 *__restrict__ tw, const DModulus &M)
{
#pragma unroll
    for (int u = 0; u < 5; u++) {
        const int s = S0 + u, half = 16 >> u;
#pragma unroll
        for (int g = 0; g < (1 << u); g++) {
            const u64 w = tw[(1u << s) + (hi << u) + (u32)g];
#pragma unroll
            for (int e = 0; e < half; e++) {
                const int j0 = (g << (5 - u)) | e, j1 = j0 | half;
                ct_bfly(x[j0], x[j1], w, M, fwd_stage_folds(s));
            }
        }
    }
}

// stages S0+4 .. S0 (inverse, Gentleman-Sande); S0 == 0 ends with the stage that carries N^-1
template <int S0>
__device__ __forceinline__ void full_inv_pass(u64 (&x)[32], u32 hi, const u64 *__restrict__ itw, const DModulus &M)
{
#pragma unroll
    for (int uu = 0; uu < 5; uu++) {
        const int u = 4 - uu, s = S0 + u, half = 16 >> u;
#pragma unroll
        for (int g = 0; g < (1 << u); g++) {
            if (s == 0) {
#pragma unroll
                for (int e = 0; e < half; e++) {
                    const int j0 = e, j1 = e | half;
                    const u64 sv = x[j0] + x[j1], d = x[j0] + (M.q << 2) - x[j1];
                    x[j0] = mulmod_lazy(M.inv_n, sv, M.delta);
                    x[j1] = mulmod_lazy(M.inv_n_w, d, M.delta);
                }
            } else {
                const u64 w = itw[(1u << s) + (hi << u) + (u32)g];
#pragma unroll
                for (int e = 0; e < half; e++) {
                    const int j0 = (g << (5 - u)) | e, j1 = j0 | half;
                    gs_bfly(x[j0], x[j1], w, M);
                }
            }
        }
    }
}

// 32 x 32 transpose between the register index and lane bits 0..4, inside the wavefront: register bit k <-> lane bit k
__device__ __forceinline__ void full_transpose(u64 (&x)[32])
{
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
    __shared__ __attribute__((aligned(16))) u64 lds[kFullLdsElems];
    const int limb = blockIdx.x;
    const int p = prime_idx ? prime_idx[limb % prime_period] : prime_base + (limb % prime_period);
    u64 *__restrict__ d = data + (long)limb * limb_stride;
    const DModulus M = mods[p];
    const u64 *__restrict__ tw = tw_all + ((size_t)p << kFullLogN);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 31;
    const u32 f = (u32)((wave << 1) | (lane >> 5)); // the 5-bit field this thread carries in exchange 1 / passes B and C
    u64 x[32], y[32];
    if (!INV) {
#pragma unroll
        for (int j = 0; j < 32; j++) x[j] = d[j * 1024 + tid]; // regs = a, thread = (b, c)
        full_fwd_pass<0>(x, 0u, tw, M);
        full_exchange(y, x, lds);                              // regs = b, thread = (a = f, c)
        full_fwd_pass<5>(y, f, tw, M);
        full_transpose(y);                                     // regs = c, lane bits 0..4 = b
        full_fwd_pass<10>(y, (f << 5) | (u32)lo, tw, M);
        full_transpose(y);                                     // regs = b, lane bits 0..4 = c
#pragma unroll
        for (int j = 0; j < 32; j++) d[(int)f * 1024 + j * 32 + lo] = canon(y[j], M);
    } else {
#pragma unroll
        for (int j = 0; j < 32; j++) x[j] = d[(int)f * 1024 + j * 32 + lo]; // regs = b, thread = (a = f, c)
        full_transpose(x);                                     // regs = c, lane bits 0..4 = b
        full_inv_pass<10>(x, (f << 5) | (u32)lo, tw, M);
        full_transpose(x);                                     // regs = b, lane bits 0..4 = c
        full_inv_pass<5>(x, f, tw, M);
        full_exchange(y, x, lds);                              // regs = a, thread = (b = f, c)
        full_inv_pass<0>(y, 0u, tw, M);
#pragma unroll
        for (int j = 0; j < 32; j++) d[j * 1024 + tid] = canon(y[j], M);
    }
}

bool ntt_full_supported(const Context &c) { return c.logN == kFullLogN; }

void launch_ntt_full(const Context &c, bool inverse, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                     int prime_period, hipStream_t s)
{
    if (count <= 0) return;
    if (prime_period <= 0) prime_period = 1 << 30;
    if (!inverse)
        hipLaunchKernelGGL(ntt_full15_kernel<false>, dim3((unsigned)count), dim3(kFullThreads), 0, s, data, limb_stride, d_prime_idx, prime_base,
                           prime_period, c.d_mods, c.d_tw);
    else
        hipLaunchKernelGGL(ntt_full15_kernel<true>, dim3((unsigned)count), dim3(kFullThreads), 0, s, data, limb_stride, d_prime_idx, prime_base,
                           prime_period, c.d_mods, c.d_itw);
}

} // namespace dacapo
