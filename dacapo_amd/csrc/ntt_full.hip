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
//               -- exchange 2 (inside the wavefront): regs b <-> lane bits c
//               pass C: stages 10..14 regs = c   thread = (a, b)   twiddles per thread, contiguous over the lanes
//               -- exchange 3 (inside the wavefront): regs c <-> lane bits b, so that the store is lane-contiguous
//     inverse   the mirror image (Gentleman-Sande stages 14..0, N^-1 merged into the last one).
// HBM accesses are 8 bytes per lane, 256- or 512-byte contiguous segments per wave instruction.
//
// Exchange 1 moves the whole limb (256 KiB) through the 160 KiB LDS in two rounds: registers whose destination is one of the first
// ten waves (20 of the 32 register indices: exactly 160 KiB) first, the other twelve after those waves have read.  A thread then
// holds at most 12 old + 32 new coefficients (88 VGPRs); the kernel is built for 128 VGPRs = 16 waves per CU = one workgroup.
//
// What bounds it (profiles/r03_ntt_full.txt): VALU issue, not HBM -- with the loads AND the stores removed the forward kernel still
// takes 796 of 874 us, and SQ_ACTIVE_INST_VALU is 88-96 % of the SIMDs' cycles at the 2.14 GHz the chip holds under this load.  Hence
// exchanges 2 and 3 go through wave-private LDS regions instead of 416 DPP / permlane instructions each (full_transpose_lds), and the
// modular multiply's two carries are fenced before their zero extension (modarith.hpp): 7 052 -> 5 457 VALU instructions per thread.
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

// ---- forward passes A and B with twiddle PAIRS (w, w 2^31 mod q) and modarith.hpp's mulmod_pair: 5 mads and one fold per multiply instead
// of 7 and two.  The product comes out below 2q, so with x folded in every butterfly -- xf = fold(x) < 2q, t = w y < 2q, (x, y) <- (xf + t,
// xf + 2q - t) -- every value stays below 4q < 2^62, mulmod_pair's operand range; the first stage's x is canonical and needs no fold.  Pass
// C then runs on words: its inputs (< 4q) are inside every range its fold schedule assumes.  Same residues as the word butterflies, hence
// the same canonical outputs, bit for bit.  Tables: Context::d_tw2 (60-bit build, N = 2^15).  Measured on 4096 limbs (profiles/
// r03_ntt_full.txt): pairs in A 784 us (words: 781), A + B 742, A + B + C 784 (pass C's 31 per-thread pairs are 16-byte loads: twice the
// twiddle bytes, what round 2 found on the tiles), B's pairs through scalar loads + select 781-796; the inverse with pairs in any prefix
// of its passes 1068-1138 against 910 on words (its first stage needs 16 pairs at once) -- so: forward A + B, inverse words.
#ifndef DC_FULL_PAIRS
#define DC_FULL_PAIRS (!DC_GENERIC_WIDTH)
#endif
#if DC_FULL_PAIRS
template <bool FOLD>
__device__ __forceinline__ void ct_bfly_p(u64 &x, u64 &y, u64 w, u64 W, const DModulus &M)
{
    const u64 xf = FOLD ? fold60(x, M.delta) : x;
    const u64 t = mulmod_pair(w, W, y, M.delta);
    x = xf + t;
    y = xf + (M.q << 1) - t;
}
template <int U>
__device__ __forceinline__ void full_stage_a_p(u64 (&x)[32], const u64 *__restrict__ tw2, const DModulus &M)
{
#pragma unroll
    for (int g = 0; g < (1 << U); g++) {
        const u64 w = tw2[2 * ((1u << U) + (u32)g)], W = tw2[2 * ((1u << U) + (u32)g) + 1]; // wave-uniform: scalar loads
#pragma unroll
        for (int e = 0; e < (16 >> U); e++) {
            const int j0 = (g << (5 - U)) | e;
            ct_bfly_p<(U != 0)>(x[j0], x[j0 | (16 >> U)], w, W, M);
        }
    }
}
__device__ __forceinline__ void full_fwd_pass_a_p(u64 (&x)[32], const u64 *__restrict__ tw2, const DModulus &M)
{
    full_stage_a_p<0>(x, tw2, M);
    full_stage_a_p<1>(x, tw2, M);
    full_stage_a_p<2>(x, tw2, M);
    full_stage_a_p<3>(x, tw2, M);
    full_stage_a_p<4>(x, tw2, M);
}
// pass B, each pair loaded (16 bytes per lane, two addresses per wave instruction) where it is used
__device__ __forceinline__ void full_fwd_pass_b_p(u64 (&x)[32], u32 hi, const u64 *__restrict__ tw2, const DModulus &M)
{
#pragma unroll
    for (int u = 0; u < 5; u++) {
#pragma unroll
        for (int g = 0; g < (1 << u); g++) {
            const ulonglong2 tp = *reinterpret_cast<const ulonglong2 *>(tw2 + 2 * ((size_t)(1u << (5 + u)) + (hi << u) + (u32)g));
            const int half = 16 >> u;
#pragma unroll
            for (int e = 0; e < half; e++) {
                const int j0 = (g << (5 - u)) | e;
                ct_bfly_p<true>(x[j0], x[j0 | half], tp.x, tp.y, M);
            }
        }
    }
}

// ---- inverse passes B and A with twiddle pairs (round 5) ------------------------------------------------------------------------------
// Gentleman-Sande on a pair: s = x + y, d = x + 2q - y, (x, y) <- (fold(s), w d) with mulmod_pair.  fold60 takes any 64-bit value to < 2q and
// the pair product is < 2q, so with inputs below 2q every value stays below 2q and d below 4q < 2^62, mulmod_pair's operand range.  The first
// pair stage follows a word pass, whose outputs are below 4q: it takes d = x + 4q - y (< 8q) through one more fold.  Same residues as
// gs_bfly, canonical stores: bit-identical results.
// Round 3 measured pairs in "any prefix" of the inverse's passes as a loss (1 068-1 138 us against 910): every prefix contains pass C, whose
// 31 per-thread pairs are 16-byte loads of a table that does not fit L2 twice.  The forward kernel's winning set {A, B} is the inverse's
// SUFFIX {B, A}: pass B's twiddles depend on the 5-bit field a half-wave carries (two addresses per wave instruction, L1 broadcasts) and
// pass A's on the register index only (scalar loads) -- the pairs' doubled bytes cost nothing there.  Loading every pair where it is used
// also drops the eight per-thread multiplications by psi^(N/2) that rebuilt the odd twiddles of pass B's widest stage.
// Table: Context::d_itw2c, per prime [1026][2]: entries 0..1023 of the inverse table as pairs, then (N^-1, .) and (N^-1 psi^-bitrev(1), .).
constexpr int kFullInvPairStride = 2 * 1026;
__device__ __forceinline__ void gs_bfly_p(u64 &x, u64 &y, u64 w, u64 W, const DModulus &M, bool after_words)
{
    const u64 s = x + y;
    u64 d = x + (M.q << (after_words ? 2 : 1)) - y;
    if (after_words) d = fold60(d, M.delta);
    x = fold60(s, M.delta);
    y = mulmod_pair(w, W, d, M.delta);
}
__device__ __forceinline__ void full_inv_pass_b_p(u64 (&x)[32], u32 hi, const u64 *__restrict__ tw2, const DModulus &M)
{
#pragma unroll
    for (int u = 4; u >= 0; u--) {
#pragma unroll
        for (int g = 0; g < (1 << u); g++) {
            const ulonglong2 tp = *reinterpret_cast<const ulonglong2 *>(tw2 + 2 * ((size_t)(1u << (5 + u)) + (hi << u) + (u32)g));
            const int half = 16 >> u;
#pragma unroll
            for (int e = 0; e < half; e++) {
                const int j0 = (g << (5 - u)) | e;
                gs_bfly_p(x[j0], x[j0 | half], tp.x, tp.y, M, u == 4);
            }
        }
    }
}
template <int U>
__device__ __forceinline__ void full_inv_stage_a_p(u64 (&x)[32], const u64 *__restrict__ tw2, const DModulus &M)
{
    if constexpr (U == 0) { // the very last inverse stage carries N^-1: both outputs are products
        const u64 n0 = tw2[2 * 1024], n1 = tw2[2 * 1024 + 1], w0 = tw2[2 * 1025], w1 = tw2[2 * 1025 + 1];
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const u64 sv = x[e] + x[e | 16], d = x[e] + (M.q << 1) - x[e | 16]; // inputs < 2q: both below 4q
            x[e] = mulmod_pair(n0, n1, sv, M.delta);
            x[e | 16] = mulmod_pair(w0, w1, d, M.delta);
        }
    } else {
#pragma unroll
        for (int g = 0; g < (1 << U); g++) {
            const u64 w = tw2[2 * ((1u << U) + (u32)g)], W = tw2[2 * ((1u << U) + (u32)g) + 1]; // wave-uniform: scalar loads
#pragma unroll
            for (int e = 0; e < (16 >> U); e++) {
                const int j0 = (g << (5 - U)) | e;
                gs_bfly_p(x[j0], x[j0 | (16 >> U)], w, W, M, false);
            }
        }
    }
}
__device__ __forceinline__ void full_inv_pass_a_p(u64 (&x)[32], const u64 *__restrict__ tw2, const DModulus &M)
{
    full_inv_stage_a_p<4>(x, tw2, M);
    full_inv_stage_a_p<3>(x, tw2, M);
    full_inv_stage_a_p<2>(x, tw2, M);
    full_inv_stage_a_p<1>(x, tw2, M);
    full_inv_stage_a_p<0>(x, tw2, M);
}
#endif

#if !DC_FULL_PAIRS
constexpr int kFullInvPairStride = 2 * 1026;
#endif

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

// The same transpose with four of its five bit swaps through LDS: the kernel is bound by VALU issue (7 052 VALU instructions per thread at
// ~87 % of the SIMDs' issue rate; profiles/r03_ntt_full.txt), and the in-register form spends 416 of them per transpose.  Register bit 4
// <-> lane bit 4 stays in registers (v_permlane16_swap, one instruction per dword); that leaves, for each half of the registers, a 16 x 16
// transpose inside every 16-lane row: 16 ds_write_b64 + 16 ds_read_b64 with all lanes active, in a wave-private region of 16 rows of 65
// words (the odd row stride spreads a read's 16 rows x 2 lane rows over all banks).  No barrier: a wave's LDS operations execute in order.
constexpr int kFullTrStride = 65, kFullTrElems = 16 * kFullTrStride; // per wave: 8 320 B, 16 waves: 133 120 B of the 160 KiB
__device__ __forceinline__ void full_transpose_lds(u64 (&x)[32], u64 *__restrict__ lds, int tid)
{
#if defined(DC_FULL_NO_TRANSPOSE) // timing experiment only (wrong results)
    return;
#endif
    const int wave = tid >> 6, lane = tid & 63;
    u64 *__restrict__ wl = lds + wave * kFullTrElems;
    u64 *__restrict__ wr = wl + lane, *__restrict__ rd = wl + (lane & 15) * kFullTrStride + (lane & 48);
#pragma unroll
    for (int j = 0; j < 16; j++) lane_swap<16>(x[j], x[j | 16]);
#pragma unroll
    for (int g = 0; g < 2; g++) {
#pragma unroll
        for (int j = 0; j < 16; j++) wr[j * kFullTrStride] = x[g * 16 + j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 16; j++) x[g * 16 + j] = rd[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

#ifndef DC_FULL_LDS_TRANSPOSE
#define DC_FULL_LDS_TRANSPOSE 1
#endif
// exchanges 2 and 3.  With the LDS form, a __syncthreads() must separate them from exchange 1 (which uses the whole LDS) on either side.
__device__ __forceinline__ void full_tr(u64 (&x)[32], u64 *__restrict__ lds, int tid)
{
#if DC_FULL_LDS_TRANSPOSE
    full_transpose_lds(x, lds, tid);
#else
    full_transpose(x);
#endif
}
__device__ __forceinline__ void full_tr_sync()
{
#if DC_FULL_LDS_TRANSPOSE
    __syncthreads();
#endif
}

// Exchange 1.  Before: thread (f = wave * 2 + (lane >> 5), c = lane & 31) holds the element whose register field is r in x[r].
// After: thread (f', c) holds in y[r'] the element that thread (r', c) had in x[f'].  (forward: f = b, r = a; inverse: f = a, r = b)
template <bool IN_LOOP = false>
__device__ __forceinline__ void full_exchange(u64 (&y)[32], const u64 (&x)[32], u64 *__restrict__ lds, int tid)
{
#if defined(DC_FULL_NO_XCHG) // timing experiment only (wrong results)
#pragma unroll
    for (int j = 0; j < 32; j++) y[j] = x[j];
    return;
#endif
    const int wave = tid >> 6, lane = tid & 63, c = lane & 31, f = (wave << 1) | (lane >> 5);
    // image "for the reader": [r' = f of the writer][reader thread] ; round 1 holds the readers of waves 0..9 (r < 20)
#pragma unroll
    for (int r = 0; r < 20; r++) lds[f * 640 + r * 32 + c] = x[r];
    __syncthreads();
    if constexpr (IN_LOOP) { // y is assigned under a branch in either round: without a definition that dominates both, its live range wraps
                             // around the caller's loop and 40-70 VGPRs are spilled at the loop header
#pragma unroll
        for (int j = 0; j < 32; j++) asm volatile("" : "=v"(y[j]));
    }
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

// One limb, forward or inverse.  IN_LOOP: called from the persistent kernel's loop (see there).
template <bool INV, bool IN_LOOP, bool PAIRS>
__device__ __forceinline__ void full_limb(u64 *__restrict__ d, const u64 *__restrict__ tw, const u64 *__restrict__ tw2, const DModulus &M,
                                          u64 *__restrict__ lds, int tid)
{
    const int wave = tid >> 6, lane = tid & 63, lo = lane & 31;
    const u32 f = (u32)((wave << 1) | (lane >> 5)); // the 5-bit field this thread carries in exchange 1 / passes B and C
    u64 x[32], y[32];
#if DC_FULL_PAIRS
    if constexpr (PAIRS && !INV) {
#pragma unroll
        for (int j = 0; j < 32; j++) x[j] = d[j * 1024 + tid]; // regs = a, thread = (b, c)
        full_fwd_pass_a_p(x, tw2, M);
        full_exchange<IN_LOOP>(y, x, lds, tid);                // regs = b, thread = (a = f, c)
        full_tr_sync();
        full_fwd_pass_b_p(y, f, tw2, M);
        full_tr(y, lds, tid);                                  // regs = c, lane bits 0..4 = b
        full_fwd_pass_simple<10>(y, (f << 5) | (u32)lo, tw, M);
        full_tr(y, lds, tid);                                  // regs = b, lane bits 0..4 = c
#pragma unroll
        for (int j = 0; j < 32; j++) d[(int)f * 1024 + j * 32 + lo] = canon(y[j], M);
        return;
    }
#endif
    FullTw t;
    const u64 im = tw[1]; // psi^(N/2) (forward table) or its inverse (inverse table): wave-uniform
#if DC_FULL_PAIRS
    if constexpr (PAIRS && INV) { // pass C on words, passes B and A on pairs
        const u32 hc = (f << 5) | (u32)lo;
        full_tw_u4<10>(t, hc, tw);
#pragma unroll
        for (int j = 0; j < 32; j++) x[j] = d[(int)f * 1024 + j * 32 + lo]; // regs = b, thread = (a = f, c)
        full_tr(x, lds, tid);                                  // regs = c, lane bits 0..4 = b
        full_pass_bc<10, true>(x, t, hc, tw, im, M);
        full_tr(x, lds, tid);                                  // regs = b, lane bits 0..4 = c
        full_inv_pass_b_p(x, f, tw2, M);
        full_tr_sync();
        full_exchange<IN_LOOP>(y, x, lds, tid);                // regs = a, thread = (b = f, c)
        full_inv_pass_a_p(y, tw2, M);
#pragma unroll
        for (int j = 0; j < 32; j++) d[j * 1024 + tid] = canon(y[j], M);
        return;
    }
#endif
    if (!INV) {
#if defined(DC_FULL_NO_LOAD) // timing experiment only (wrong results)
#pragma unroll
        for (int j = 0; j < 32; j++) x[j] = (u64)(tid * 33 + j) + M.q;
#else
#pragma unroll
        for (int j = 0; j < 32; j++) x[j] = d[j * 1024 + tid]; // regs = a, thread = (b, c)
#endif
        full_pass_a<false>(x, tw, M);
        full_exchange<IN_LOOP>(y, x, lds, tid);                // regs = b, thread = (a = f, c)
        full_tr_sync();
#if defined(DC_FULL_FWD_PIPELINED)
        full_tw_small<5>(t, f, tw);
        full_pass_bc<5, false>(y, t, f, tw, im, M);
        full_tr(y, lds, tid);                                  // regs = c, lane bits 0..4 = b
        const u32 hc = (f << 5) | (u32)lo;
        full_tw_small<10>(t, hc, tw);
        full_pass_bc<10, false>(y, t, hc, tw, im, M);
#else
        full_fwd_pass_simple<5>(y, f, tw, M);
        full_tr(y, lds, tid);                                  // regs = c, lane bits 0..4 = b
        full_fwd_pass_simple<10>(y, (f << 5) | (u32)lo, tw, M);
#endif
        full_tr(y, lds, tid);                                  // regs = b, lane bits 0..4 = c
#if defined(DC_FULL_NO_STORE) // timing experiment only (wrong results): one store per thread instead of 32
        u64 acc = 0;
#pragma unroll
        for (int j = 0; j < 32; j++) acc ^= canon(y[j], M);
        d[tid] = acc;
#else
#pragma unroll
        for (int j = 0; j < 32; j++) d[(int)f * 1024 + j * 32 + lo] = canon(y[j], M);
#endif
    } else {
        const u32 hc = (f << 5) | (u32)lo;
        full_tw_u4<10>(t, hc, tw);
#pragma unroll
        for (int j = 0; j < 32; j++) x[j] = d[(int)f * 1024 + j * 32 + lo]; // regs = b, thread = (a = f, c)
        full_tr(x, lds, tid);                                  // regs = c, lane bits 0..4 = b
        full_pass_bc<10, true>(x, t, hc, tw, im, M);
        full_tw_u4<5>(t, f, tw);
        full_tr(x, lds, tid);                                  // regs = b, lane bits 0..4 = c
        full_pass_bc<5, true>(x, t, f, tw, im, M);
        full_tr_sync();
        full_exchange<IN_LOOP>(y, x, lds, tid);                // regs = a, thread = (b = f, c)
        full_pass_a<true>(y, tw, M);
#pragma unroll
        for (int j = 0; j < 32; j++) d[j * 1024 + tid] = canon(y[j], M);
    }
}

// grid = count (one limb per workgroup) or fewer: a PERSISTENT grid of one workgroup per CU, each walking over the limbs
// (limb = blockIdx.x + k * gridDim.x), so that a limb's stores drain under the next limb's loads and no workgroup is relaunched in
// between.  Two things keep hipcc from spilling in the loop form (round 3's first attempt spilled 71-94 VGPRs and ran at 1057-1128 us
// against 894): the thread index is laundered per iteration (otherwise the 62 twiddle offsets of passes B and C are hoisted out of the
// loop as invariants), and full_exchange<true> defines y[] before the two rounds' branches assign it.
template <bool INV, bool PAIRS>
__global__ __launch_bounds__(kFullThreads) void ntt_full15_kernel(u64 *__restrict__ data, long limb_stride, const int *__restrict__ prime_idx,
                                                                   int prime_base, int prime_period, const DModulus *__restrict__ mods,
                                                                   const u64 *__restrict__ tw_all, const u64 *__restrict__ tw2_all, int count)
{
    __shared__ __attribute__((aligned(16))) u64 lds[kFullLdsElems];
    // The grid walks over the limbs PRIME BY PRIME (all limbs of residue 0 mod the period, then residue 1, ...): the workgroups of an XCD
    // then read one or two primes' twiddle tables at a time (0.25 MB each) out of its 4 MB L2 instead of all of them.
    const int period = prime_period < count ? prime_period : count; // residues that occur
    const int big = count % period, ns = count / period, nb = ns + 1; // the first `big` residues have nb limbs, the others ns
    for (int pos = blockIdx.x; pos < count; pos += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid)); // a per-iteration value as far as the optimiser can tell
        int r, k;
        if (pos < big * nb)
            r = pos / nb, k = pos - r * nb;
        else {
            const int q = pos - big * nb;
            r = big + q / ns, k = q - (q / ns) * ns;
        }
        r = __builtin_amdgcn_readfirstlane(r), k = __builtin_amdgcn_readfirstlane(k); // (the divisions run on the vector unit: back to scalars)
        const int limb = k * period + r;
        const int p = prime_idx ? prime_idx[r] : prime_base + r;
        const DModulus M = mods[p];
        full_limb<INV, true, PAIRS>(data + (long)limb * limb_stride, tw_all + ((size_t)p << kFullLogN),
                                    PAIRS ? tw2_all + (INV ? (size_t)p * kFullInvPairStride : (size_t)p << (kFullLogN + 1)) : nullptr, M, lds, tid);
        __syncthreads(); // the next limb's exchange / transposes write what the slowest waves may still be reading
    }
}

bool ntt_full_supported(const Context &c) { return c.logN == kFullLogN; }
long ntt_full_min_limbs(bool inverse)
{ // Below this many limbs the two-launch tiles (16 workgroups per limb, several per CU) are faster.  Measured on the final kernels
  // (profiles/r03_ntt_full_check.txt and r03_ntt_full.txt), single-crossing vs two-launch: forward 127 vs 119 us at 512 limbs, 150 vs 157
  // at 640, 159 vs 193 at 768, 216 vs 266 at 1024, 415 vs 587 at 2048, 753 vs 1123 at 4096; inverse 172 vs 149 at 640, 230 vs 204 at 896,
  // 236 vs 244 at 1024, 480 vs 512 at 2048, 919 vs 997 at 4096.  0 = never.
    return (long)option(inverse ? OPT_NTT_FULL_INV_MIN_LIMBS : OPT_NTT_FULL_MIN_LIMBS);
}

static int full_persist_grid(bool inverse)
{ // options ntt_full_persist / ntt_full_inv_persist: workgroups of the persistent grid (-1, the default: one per CU); 0 = one workgroup
  // per limb.  Measured on 4096 limbs: forward 788 -> 782 us, inverse 896 -> 901 (a tie: both directions take the same form).
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    const int f = (int)option(OPT_NTT_FULL_PERSIST), i = (int)option(OPT_NTT_FULL_INV_PERSIST);
    const int g = inverse && i >= 0 ? i : f;
    return g < 0 ? cus : g;
}

// Does the single-crossing kernel beat the two-launch tiles for this launch?  Its persistent grid runs ceil(count / CUs) rounds of one limb
// per CU, so a launch just above a multiple of the CU count pays a whole round for a few limbs, while the tiles (16 workgroups per limb)
// scale smoothly -- and got faster in round 4 (COLS placement).  Measured on the round-5 kernels (profiles/r05_ntt_full_check.txt; single
// crossing / tiles, us): forward 640 limbs 166 / 138, 768 176 / 177, 1 024 214 / 244, 1 536 321 / 438, 2 048 410 / 580; inverse (pairs in
// passes B and A) 640 163 / 140, 768 170 / 169, 1 024 219 / 222, 1 536 340 / 364, 2 048 448 / 480, 4 096 850 / 939.  So: from the option's
// limb count on, when the last round is full enough -- or always from 2 048 limbs on, where the tiles' second crossing leaves the caches.
bool ntt_full_pays(bool inverse, int count)
{
    const long m = ntt_full_min_limbs(inverse);
    if (m <= 0 || count < m) return false;
    if (count >= 2048) return true;
    const int pg = full_persist_grid(inverse);
    if (pg <= 0) return true; // (one workgroup per limb: no rounds)
    const long rounds = (count + pg - 1) / pg;
    // (forward: the single-crossing kernel wins from 768 limbs on whatever the last round holds -- 900 limbs 205 / 223 us, 1 300 298 / 368;
    //  inverse: only when the last round is full -- 900 limbs 214 / 199, 1 300 318 / 315)
    return !inverse || (long)count * 100 >= rounds * pg * 96;
}

static bool full_pairs()
{ // option ntt_full_pairs = 0: word butterflies in every forward pass (A/B measurements; the pair tables exist either way)
    return option(OPT_NTT_FULL_PAIRS) != 0;
}

void launch_ntt_full(const Context &c, bool inverse, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                     int prime_period, hipStream_t s)
{
    if (count <= 0) return;
    if (prime_period <= 0) prime_period = 1 << 30;
    const int pg = full_persist_grid(inverse);
    const unsigned grid = (unsigned)(pg > 0 && count > pg ? pg : count);
#if DC_FULL_PAIRS
    if (!inverse && full_pairs() && c.d_tw2) {
        DC_LAUNCH((ntt_full15_kernel<false, true>), dim3(grid), dim3(kFullThreads), 0, s, data, limb_stride, d_prime_idx, prime_base,
                           prime_period, c.d_mods, c.d_tw, c.d_tw2, count);
        return;
    }
#endif
#if DC_FULL_PAIRS
    if (inverse && option(OPT_NTT_FULL_INV_PAIRS) != 0 && c.d_itw2c) {
        DC_LAUNCH((ntt_full15_kernel<true, true>), dim3(grid), dim3(kFullThreads), 0, s, data, limb_stride, d_prime_idx, prime_base,
                           prime_period, c.d_mods, c.d_itw, c.d_itw2c, count);
        return;
    }
#endif
    if (!inverse)
        DC_LAUNCH((ntt_full15_kernel<false, false>), dim3(grid), dim3(kFullThreads), 0, s, data, limb_stride, d_prime_idx, prime_base,
                           prime_period, c.d_mods, c.d_tw, (const u64 *)nullptr, count);
    else
        DC_LAUNCH((ntt_full15_kernel<true, false>), dim3(grid), dim3(kFullThreads), 0, s, data, limb_stride, d_prime_idx, prime_base,
                           prime_period, c.d_mods, c.d_itw, (const u64 *)nullptr, count);
}

} // namespace dacapo
