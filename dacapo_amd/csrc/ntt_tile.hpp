// One workgroup (256 threads) = one tile of one RNS limb; the negacyclic NTT of a limb of N = 2^logN coefficients is
// two launches ("phases") of this tile routine, so that a single limb already spreads over many workgroups and a key
// switch at small level still fills the chip:
//
//   forward  (SEAL ntt_negacyclic_harvey: natural in, bit-reversed out; stage with m groups uses psi^bitrev(m+i))
//     phase COLS : first k1 stages.  View the limb as an [N1 = 2^k1][N2 = N/N1] matrix; these stages only
//                  couple elements of one column, and every column uses the same twiddles tw[2^s + g].
//                  Tile = B adjacent columns x N1 rows (row segments of B*8 bytes are contiguous in HBM).
//     phase ROWS : last k2 = logN - k1 stages.  They only couple elements of one row (N2 contiguous
//                  coefficients); row r, local stage s, local group g uses tw[((N1 + r) << s) + g].
//                  Tile = B adjacent rows.
//   inverse (Gentleman-Sande, bit-reversed in, natural out): ROWS phase first, then COLS, same tiles and the
//     same table indices with inverse twiddles; N^{-1} is merged into the very last stage.
//
// Three geometries (template parameter LOGE = log2 of the coefficients a thread keeps in registers):
//   LOGE = 3 : 2048-coefficient tiles, radix-8 passes (3 stages per pass)  -- throughput: fewest LDS exchanges
//   LOGE = 2 : 1024-coefficient tiles, radix-4 passes (2 stages per pass)  -- latency: twice the workgroups, half
//              the butterflies per thread; used when a launch has too few tiles to occupy the chip (a lone wave per
//              SIMD is bound by its own instruction latency, so halving its work nearly halves the phase).
//   LOGE = 1 :  512-coefficient tiles, radix-2 passes, ONE butterfly per thread and stage -- single-ciphertext steps, where
//              even the radix-4 tiles leave half the SIMDs without a wave.  A pass boundary is then a 2 x 2 transpose between
//              a register pair and one lane bit: when the partner lane sits in the same wavefront (distance < 64) it is a
//              lane_swap (lane_xchg.hpp: v_permlane32/16_swap, masked DPP moves) -- no LDS, no barrier; only the first one
//              or two boundaries of a tile cross wavefronts and go through LDS.
// Between the other passes the tile is transposed through LDS.  The LDS image is laid out for the READER: element j of
// thread t lives at j*(T+pad)+t, so every ds_read_b64 is lane-contiguous.
#pragma once
#include <stdlib.h>

#include "lane_xchg.hpp"
#include "modarith.hpp"
#include "options.hpp"

namespace dacapo {

constexpr int kTileThreads = 256;
#ifndef DC_LDS_PAD
#define DC_LDS_PAD 4
#endif
constexpr int kLdsPad = DC_LDS_PAD; // u64 words between the LDS rows of consecutive register indices (measured: profiles/r02_lds_pad_sweep.txt)
template <int LOGE>
struct TileGeo {
    static constexpr int E = 1 << LOGE;            // coefficients per thread
    static constexpr int LOG = 8 + LOGE;           // log2(coefficients per tile)
    static constexpr int ELEMS = 1 << LOG;
    static constexpr int LDS_ELEMS = E * (kTileThreads + kLdsPad);
};
// the throughput geometry's constants, used by host code to size grids
constexpr long kSmallTileWgsN16 = 5000; // (N = 2^16: config 3's ring)
constexpr int kTileLog = TileGeo<3>::LOG;
constexpr int kTileElems = TileGeo<3>::ELEMS;
constexpr int kTileLdsElems = TileGeo<3>::LDS_ELEMS;

template <int LOGE>
__host__ __device__ constexpr int pass_stages(int K, int p) { return (K - LOGE * p) >= LOGE ? LOGE : (K - LOGE * p); }
template <int LOGE>
__host__ __device__ constexpr int num_passes(int K) { return (K + LOGE - 1) / LOGE; }

// ---- per-pass index algebra (all compile-time foldable once loops are unrolled) ----------------------
template <int K, int LOGE>
struct PassMap {
    // thread sub-index s in [0, n/E), register j in [0,E)  ->  local coefficient index in [0, n)
    __device__ static __forceinline__ int idx_of(int p, int s, int j)
    {
        const int s0 = LOGE * p, r = pass_stages<LOGE>(K, p);
        const int u = j >> r, kk = j & ((1 << r) - 1);
        const int vt = (s << (LOGE - r)) | u;
        const int lob = K - s0 - r;
        const int hi = vt >> lob, lo = vt & ((1 << lob) - 1);
        return (hi << (K - s0)) | (kk << lob) | lo;
    }
    // inverse of idx_of: local index -> (s, j) under pass p
    __device__ static __forceinline__ void sj_of(int p, int idx, int &s, int &j)
    {
        const int s0 = LOGE * p, r = pass_stages<LOGE>(K, p);
        const int lob = K - s0 - r;
        const int lo = idx & ((1 << lob) - 1);
        const int kk = (idx >> lob) & ((1 << r) - 1);
        const int hi = idx >> (K - s0);
        const int vt = (hi << lob) | lo;
        s = vt >> (LOGE - r);
        j = ((vt & ((1 << (LOGE - r)) - 1)) << r) | kk;
    }
};

// Lazy-reduction schedule.  mulmod_lazy(w, y) takes any w < 2^60 and y < 1.875 * 2^63 (its middle term a0*b1 + a1*b0 + carry
// word then stays below 2^64 and the product below 2^124) and returns a value < 2^62; fold60 takes any 64-bit value.  So a
// butterfly's "x" input only has to be folded often enough for the sums to fit 64 bits and the next multiplicand to stay under
// 1.875 * 2^63 = 3.75 * 2^62:
//   forward (Cooley-Tukey):  F stage  x' , y' = fold60(x) +- w y      -> every output < 1.25 * 2^62 (+2^32)
//                            N stage  x' , y' = x +- w y (no fold)     -> bound grows by 2^62 per stage: 2.25, 3.25 (* 2^62) < 2^64
//                            schedule F N N F N N ... from the first stage of every phase (inputs from memory are canonical or
//                            a previous phase's outputs, < 3.25 * 2^62: fine for an F stage)
//   inverse (Gentleman-Sande): x' = fold60(x + y), y' = w (x + 4q - y) at every stage.  The subtraction needs 4q > y, so every
//                            value must stay below 4q: the sum is folded each time (an unfolded x + y could be the next
//                            stage's subtrahend, and a larger multiple of q in its place would push the multiplicand past
//                            its bound).  Forward has no such constraint: its subtrahend is always the fresh product w y < 4q.
// Every value stored as a RESULT is canonical; intermediate (phase-to-phase) limbs may hold any of the lazy ranges above.
__device__ __forceinline__ constexpr bool fwd_stage_folds(int stage_in_phase) { return stage_in_phase % 3 == 0; }

// Cooley-Tukey butterfly:  (x, y) -> (x + w y, x - w y)
__device__ __forceinline__ void ct_bfly(u64 &x, u64 &y, u64 w, const DModulus &M, bool fold)
{
    const u64 xf = fold ? fold60(x, M.delta) : x;
    const u64 t = mulmod_lazy(w, y, M.delta); // < 4q < 2^62
    x = xf + t;
    y = xf + (M.q << 2) - t;                  // 4q > t
}
#if !DC_GENERIC_WIDTH
// Cooley-Tukey butterfly on a twiddle PAIR (w, W = w 2^31 mod q) and modarith.hpp's mulmod_pair: 5 mads and one fold per multiply instead of 7
// and two.  The product comes out below 2q, so with x folded in every butterfly -- xf = fold(x) < 2q, t = w y < 2q, (x, y) <- (xf + t,
// xf + 2q - t) -- every value stays below 4q < 2^62, mulmod_pair's operand range; a stage whose x is canonical needs no fold (x + t < 3q).
// Same residues as ct_bfly, hence the same canonical results, bit for bit.
__device__ __forceinline__ void ct_bfly_pair(u64 &x, u64 &y, u64 w, u64 W, const DModulus &M, bool fold)
{
    const u64 xf = fold ? fold60(x, M.delta) : x;
    const u64 t = mulmod_pair(w, W, y, M.delta);
    x = xf + t;
    y = xf + (M.q << 1) - t;
}
#endif
// Gentleman-Sande butterfly, values < 4q (< 2^62) in and out:  (x, y) -> (x + y, (x - y) w)
// (every value here is a canonical input, a fold60 result or a mulmod_lazy result, all < 4q = 2^62 - 4 delta)
__device__ __forceinline__ void gs_bfly(u64 &x, u64 &y, u64 w, const DModulus &M)
{
    const u64 s = x + y;                      // < 2^63
    const u64 d = x + (M.q << 2) - y;         // 4q > y ; < 2^63
    x = fold60(s, M.delta);
    y = mulmod_lazy(w, d, M.delta);           // < 4q
}

// XCD-aware placement of COLS tiles.  A COLS tile touches row segments of B * 8 bytes (64 for the radix-8 tiles of a 256-row phase, 32 / 16
// for the latency geometries) while the memory side moves 128-byte lines, so 16 / B neighbouring tiles share every line they touch; workgroup w
// of a launch runs on XCD w mod 8 (tools/experiments/xcc_probe.hip), each XCD with its own L2 -- in launch order the sharers sat on different
// XCDs and every line was fetched 16 / B times (measured: the COLS phases read 2.0x their limbs, profiles/r04_hybrid_ks_kernels.txt).  The map
// below hands the tiles of one line to workgroups w, w + 8, ...: same XCD, dispatched together, the later ones hit the first one's line.
// A bijection on the tiles of a limb; every address of a COLS tile comes from ntt_tile_x / tile_gidx, which both apply it.
template <int K, int LOGE>
__device__ __forceinline__ int cols_tile_of(int bx, int logN)
{
    constexpr int LOGB = TileGeo<LOGE>::LOG - K;
    constexpr int LOGS = LOGB >= 4 ? 0 : 4 - LOGB; // log2(tiles per 128-byte line)
#if defined(DC_EXP_NO_COLS_REMAP)
    return bx;
#endif
    if (LOGS == 0 || logN - TileGeo<LOGE>::LOG < 3 + LOGS) return bx; // (fewer than 8 line groups in a limb: launch order)
    const int r = bx & ((8 << LOGS) - 1);
    return (bx & ~((8 << LOGS) - 1)) | ((r & 7) << LOGS) | (r >> 3);
}

// Ld: u64 operator()(int gidx)            coefficient gidx (0..N-1) of this limb: canonical, or what a previous phase stored
// St: void operator()(int gidx, u64 v)    v canonical if CANON else lazy (any of the ranges of the schedule above)
// PRELOADED: x[] already holds the first pass's coefficients (register j <-> idx_of(first pass, s, j)); ld unused.
// KEEP     : leave the result in x[] (canonical if CANON) instead of calling st.  The last pass of an inverse phase and
//            the first pass of a forward phase of the same shape use the same thread<->coefficient map, so an inverse
//            tile can hand its output to a forward tile in registers (the fused iNTT -> base change -> NTT kernels).
// wext (latency geometries only): the tile's twiddles as tile_twiddles() fetched them -- a kernel that runs the same tile of the same prime
// several times (the digits of a key switch) fetches them once, not once per transform behind the exchanges' memory fences.
// PAIRS (forward COLS tiles of the 60-bit build): `tw` is the prime's table of twiddle PAIRS (Context::d_twc2: the first 2^k1 entries of the
// forward table, 16 bytes each: a COLS phase's twiddles depend on the row group only -- a table of 2^K entries shared by every column of every
// tile, L1 / L2 resident -- so the pairs' doubled bytes cost nothing, unlike a ROWS phase's per-row tables) and the butterflies are
// ct_bfly_pair: ~13 % fewer VALU instructions per butterfly.  The phase's inputs must be canonical (a forward transform's first phase: they
// are); its outputs are below 4q, inside every range the ROWS phase's schedule assumes.
// (Round 5, measured and not kept: pairs in the generic forward ROWS phase at N = 2^17 -- a 16-byte-per-twiddle copy of the whole table, both
// operands folded in the first stage -- with rows_prime_major supplying L2 hits for the doubled table bytes: config 4 2.789 / 2.798 s against
// 2.797 / 2.767 without, a single hop at 31 primes 475 against 466 us; profiles/r05_experiments.txt item 21.)
// FOLD0 (PAIRS only): fold x in stage 0 as well -- for callers whose inputs are residues below 4q but not necessarily canonical
template <int K, int LOGE, bool COLS, bool INV, bool CANON, bool PRELOADED, bool KEEP, bool PAIRS, bool FOLD0 = false, class Ld, class St>
__device__ __forceinline__ void ntt_tile_core(u64 (&x)[1 << LOGE], const DModulus M, const u64 *__restrict__ tw, int logN, int tile,
                                              Ld ld, St st, u64 *__restrict__ lds, const u64 (*wext)[1 << LOGE] = nullptr)
{
    static_assert(!PAIRS || (COLS && !INV && !DC_GENERIC_WIDTH), "twiddle pairs: forward COLS tiles of the 60-bit build");
    constexpr int E = 1 << LOGE, n = 1 << K, LOGB = TileGeo<LOGE>::LOG - K, B = 1 << LOGB, T = kTileThreads, SUBT = n / E;
    constexpr int NP = num_passes<LOGE>(K);
    constexpr int stride = T + kLdsPad;
    static_assert(LOGB >= 0, "sub-transform larger than the tile");
    // ROWS tiles give each sub-transform SUBT consecutive threads: when that is at most one wavefront, an exchange only moves
    // data between lanes of the same wave, LDS serves a wave's requests in order, and the workgroup barrier can be replaced by
    // "my own LDS traffic has drained" (no s_barrier, no waiting for the other three waves).
    constexpr bool WAVE_LOCAL = !COLS && SUBT <= 64;
    auto exchange_sync = [&]() {
        if (WAVE_LOCAL)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else
            __syncthreads();
    };
    const int t = threadIdx.x;
    int b, s;
    if (COLS) {
        b = t & (B - 1);
        s = t >> LOGB;
    } else {
        s = t & (SUBT - 1);
        b = t / SUBT;
    }
    const int sh = logN - K; // COLS: log2(column stride) ; ROWS: log2(N1)
    if (COLS) tile = cols_tile_of<K, LOGE>(tile, logN);
    const int lane_id = tile * B + b;
    const u32 twroot = COLS ? 1u : ((1u << sh) + (u32)lane_id);
    auto gidx = [&](int idx) -> int { return COLS ? ((idx << sh) + lane_id) : ((lane_id << K) + idx); };
    // Fetch every pass's twiddles (<= E - 1 per pass) before the first butterfly.  The barriers between
    // passes pin memory operations in place, so otherwise each pass starts by waiting for its own twiddle loads.
    // The radix-8 ROWS tiles too (round 4): their late stages' twiddles are one load per lane and butterfly, and issued pass by pass they were
    // waited for -- 160 limbs at N = 2^17: inverse ROWS 98 -> 82 us, forward ROWS 87 -> 80 us (44 / 50 -> 65 / 64 VGPRs).  A radix-8 COLS tile's
    // twiddles are a 256-entry table shared by its columns: cache hits, left where they are used.
    constexpr bool PF = (LOGE <= 2) || !COLS; // (radix-8 COLS tiles with the prefetch: 69.8 -> 67.7 us forward, 83.7 -> 82.2 inverse: not worth their fused forms' registers)
    u64 wpre[PF ? NP : 1][E];
    u64 wpre2[(PF && PAIRS) ? NP : 1][E]; // the pairs' second words
    if (PF && !wext) {
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
            const int p = INV ? (NP - 1 - pp) : pp;
            const int s0 = LOGE * p, r = pass_stages<LOGE>(K, p);
#pragma unroll
            for (int u = 0; u < (1 << (LOGE - r)); u++) {
                const int vt = (s << (LOGE - r)) | u;
                const int hi = vt >> (K - s0 - r);
#pragma unroll
                for (int st_ = 0; st_ < r; st_++) {
#pragma unroll
                    for (int g = 0; g < (1 << st_); g++) {
                        const u32 ti = (twroot << (s0 + st_)) + (u32)((hi << st_) | g);
                        if constexpr (PAIRS) {
                            const ulonglong2 tp = *reinterpret_cast<const ulonglong2 *>(tw + 2 * (size_t)ti);
                            wpre[pp][(u << r) | ((1 << st_) + g)] = tp.x, wpre2[PF ? pp : 0][(u << r) | ((1 << st_) + g)] = tp.y;
                        } else
                            wpre[pp][(u << r) | ((1 << st_) + g)] = tw[ti];
                    }
                }
            }
        }
    }

    bool lds_dirty = false, from_lds = false; // compile-time after unrolling: an LDS image exists / the next pass reads it
#pragma unroll
    for (int pp = 0; pp < NP; pp++) {
        const int p = INV ? (NP - 1 - pp) : pp;
        const int s0 = LOGE * p, r = pass_stages<LOGE>(K, p);
        const bool first = (pp == 0), last = (pp == NP - 1);
        // ---- load
        if (first) {
            if (!PRELOADED) {
#pragma unroll
                for (int j = 0; j < E; j++) x[j] = ld(gidx(PassMap<K, LOGE>::idx_of(p, s, j)));
            }
        } else if (from_lds) {
#pragma unroll
            for (int j = 0; j < E; j++) x[j] = lds[j * stride + t];
            from_lds = false;
        }
        // ---- butterflies: 2^(LOGE-r) independent radix-2^r networks per thread
#pragma unroll
        for (int u = 0; u < (1 << (LOGE - r)); u++) {
            const int vt = (s << (LOGE - r)) | u;
            const int hi = vt >> (K - s0 - r);
#pragma unroll
            for (int tt = 0; tt < r; tt++) {
                const int st_ = INV ? (r - 1 - tt) : tt; // stage within the pass
                const int gs = s0 + st_;                 // local stage index
                const int half = 1 << (r - 1 - st_);
#pragma unroll
                for (int g = 0; g < (1 << st_); g++) {
                    const u32 twi = (twroot << gs) + (u32)((hi << st_) | g);
#if defined(DC_EXP_NO_TWLOAD) // timing experiment only (wrong results; profiles/r02_experiments.txt): twiddles from registers, not memory
                    const u64 w = M.inv_n + twi;
#else
                    u64 w, W2 = 0;
                    if constexpr (PAIRS) {
                        if (PF)
                            w = wpre[PF ? pp : 0][(u << r) | ((1 << st_) + g)], W2 = wpre2[PF ? pp : 0][(u << r) | ((1 << st_) + g)];
                        else {
                            const ulonglong2 tp = *reinterpret_cast<const ulonglong2 *>(tw + 2 * (size_t)twi);
                            w = tp.x, W2 = tp.y;
                        }
                    } else
                        w = PF ? (wext ? wext[PF ? pp : 0][(u << r) | ((1 << st_) + g)] : wpre[PF ? pp : 0][(u << r) | ((1 << st_) + g)]) : tw[twi];
#endif
#pragma unroll
#if defined(DC_EXP_NO_BFLY) // timing experiment only (wrong results): loads, exchanges and stores without the arithmetic
                    for (int e = 0; e < 0; e++) {
#else
                    for (int e = 0; e < half; e++) {
#endif
                        const int j0 = (u << r) | (g << (r - st_)) | e, j1 = j0 | half;
                        if constexpr (PAIRS) {
#if !DC_GENERIC_WIDTH
                            ct_bfly_pair(x[j0], x[j1], w, W2, M, gs != 0 || FOLD0); // (stage 0 of a forward transform: canonical inputs unless FOLD0)
#endif
                        } else if (!INV)
                            ct_bfly(x[j0], x[j1], w, M, fwd_stage_folds(gs));
                        else if (COLS && gs == 0) { // very last inverse stage: fold N^{-1} in
                            u64 sv = x[j0] + x[j1];
                            u64 d = x[j0] + (M.q << 2) - x[j1];
                            x[j0] = mulmod_lazy(M.inv_n, sv, M.delta);
                            x[j1] = mulmod_lazy(M.inv_n_w, d, M.delta);
                        } else
                            gs_bfly(x[j0], x[j1], w, M);
                    }
                }
            }
        }
        // ---- store
        if (last) {
            if (KEEP) {
                if (CANON) {
#pragma unroll
                    for (int j = 0; j < E; j++) x[j] = canon(x[j], M);
                }
            } else {
#pragma unroll
                for (int j = 0; j < E; j++) st(gidx(PassMap<K, LOGE>::idx_of(p, s, j)), CANON ? canon(x[j], M) : x[j]);
            }
        } else {
            const int pn = INV ? p - 1 : p + 1;
            // LOGE = 1: the boundary between passes q and q + 1 exchanges register bit and thread-index bit K - q - 2, i.e. lanes
            // at distance 2^(K-q-2) (times the column interleave of a COLS tile)
            const int lane_dist = LOGE == 1 ? ((1 << (K - (INV ? pn : p) - 2)) << (COLS ? LOGB : 0)) : 64;
            if (LOGE == 1 && lane_dist < 64) {
                if constexpr (LOGE == 1) { // (x[] has two elements only in this instantiation)
                    switch (lane_dist) { // a constant once the pass loop is unrolled
                    case 32: lane_swap<32>(x[0], x[1]); break;
                    case 16: lane_swap<16>(x[0], x[1]); break;
                    case 8: lane_swap<8>(x[0], x[1]); break;
                    case 4: lane_swap<4>(x[0], x[1]); break;
                    case 2: lane_swap<2>(x[0], x[1]); break;
                    default: lane_swap<1>(x[0], x[1]); break;
                    }
                }
            } else {
                if (lds_dirty) exchange_sync(); // everyone has finished reading the previous image
#pragma unroll
                for (int j = 0; j < E; j++) {
                    int s2, j2;
                    PassMap<K, LOGE>::sj_of(pn, PassMap<K, LOGE>::idx_of(p, s, j), s2, j2);
                    const int t2 = COLS ? ((s2 << LOGB) | b) : (b * SUBT + s2);
                    lds[j2 * stride + t2] = x[j];
                }
                exchange_sync();
                lds_dirty = true, from_lds = true;
            }
        }
    }
}

template <int K, int LOGE, bool COLS, bool INV, bool CANON, bool PRELOADED, bool KEEP, class Ld, class St>
__device__ __forceinline__ void ntt_tile_x(u64 (&x)[1 << LOGE], const DModulus M, const u64 *__restrict__ tw, int logN, int tile,
                                           Ld ld, St st, u64 *__restrict__ lds, const u64 (*wext)[1 << LOGE] = nullptr)
{
    ntt_tile_core<K, LOGE, COLS, INV, CANON, PRELOADED, KEEP, false>(x, M, tw, logN, tile, ld, st, lds, wext);
}
// forward COLS tile; tw2 = the prime's pair table when the context has one (60-bit build), else nullptr and `tw` is used on words.
// (the choice is uniform over the launch; both branches are instantiated)
template <int K, int LOGE, bool CANON, bool PRELOADED, bool KEEP, bool FOLD0 = false, class Ld, class St>
__device__ __forceinline__ void ntt_tile_fcols(u64 (&x)[1 << LOGE], const DModulus M, const u64 *__restrict__ tw, const u64 *__restrict__ tw2,
                                               int logN, int tile, Ld ld, St st, u64 *__restrict__ lds)
{
#if !DC_GENERIC_WIDTH
    if (tw2) {
        ntt_tile_core<K, LOGE, true, false, CANON, PRELOADED, KEEP, true, FOLD0>(x, M, tw2, logN, tile, ld, st, lds);
        return;
    }
#endif
    ntt_tile_core<K, LOGE, true, false, CANON, PRELOADED, KEEP, false>(x, M, tw, logN, tile, ld, st, lds);
}

// global coefficient index of register j of this thread under pass p (what ld/st are called with): lets a kernel fill
// x[] itself (PRELOADED) or consume it (KEEP) with all its loads in flight at once.  An inverse tile starts and a forward
// tile ends with pass num_passes - 1.
template <int K, int LOGE, bool COLS>
__device__ __forceinline__ int tile_gidx(int p, int logN, int tile, int j)
{
    constexpr int LOGB = TileGeo<LOGE>::LOG - K, B = 1 << LOGB, SUBT = (1 << K) >> LOGE;
    const int t = threadIdx.x;
    const int b = COLS ? (t & (B - 1)) : (t / SUBT), s = COLS ? (t >> LOGB) : (t & (SUBT - 1));
    if (COLS) tile = cols_tile_of<K, LOGE>(tile, logN);
    const int lane_id = tile * B + b, idx = PassMap<K, LOGE>::idx_of(p, s, j);
    return COLS ? ((idx << (logN - K)) + lane_id) : ((lane_id << K) + idx);
}

// every pass's twiddles of this thread, in the layout ntt_tile_x keeps them in (its `wext` argument); LOGE <= 2
template <int K, int LOGE, bool COLS, bool INV>
__device__ __forceinline__ void tile_twiddles(u64 (&w)[num_passes<LOGE>(K)][1 << LOGE], const u64 *__restrict__ tw, int logN, int tile)
{
    static_assert(LOGE <= 2, "the radix-8 tiles read their twiddles pass by pass");
    constexpr int n = 1 << K, LOGB = TileGeo<LOGE>::LOG - K, B = 1 << LOGB, SUBT = n >> LOGE, NP = num_passes<LOGE>(K);
    const int t = threadIdx.x;
    const int b = COLS ? (t & (B - 1)) : (t / SUBT), s = COLS ? (t >> LOGB) : (t & (SUBT - 1));
    if (COLS) tile = cols_tile_of<K, LOGE>(tile, logN);
    const u32 twroot = COLS ? 1u : ((1u << (logN - K)) + (u32)(tile * B + b));
#pragma unroll
    for (int pp = 0; pp < NP; pp++) {
        const int p = INV ? (NP - 1 - pp) : pp;
        const int s0 = LOGE * p, r = pass_stages<LOGE>(K, p);
#pragma unroll
        for (int u = 0; u < (1 << (LOGE - r)); u++) {
            const int vt = (s << (LOGE - r)) | u;
            const int hi = vt >> (K - s0 - r);
#pragma unroll
            for (int st_ = 0; st_ < r; st_++) {
#pragma unroll
                for (int g = 0; g < (1 << st_); g++) w[pp][(u << r) | ((1 << st_) + g)] = tw[(twroot << (s0 + st_)) + (u32)((hi << st_) | g)];
            }
        }
    }
}

template <int K, int LOGE, bool COLS, bool INV, bool CANON, class Ld, class St>
__device__ __forceinline__ void ntt_tile(const DModulus M, const u64 *__restrict__ tw, int logN, int tile, Ld ld, St st,
                                         u64 *__restrict__ lds)
{
    u64 x[1 << LOGE];
    ntt_tile_x<K, LOGE, COLS, INV, CANON, false, false>(x, M, tw, logN, tile, ld, st, lds);
}

// geometry choice for a launch of `limbs` limb-phases: the latency geometry while the throughput one would leave most
// of the 256 CUs without a workgroup, the one-butterfly geometry while even that leaves SIMDs without a wave
// The launch-shape table, keyed by the ring: launches below this many 2048-coefficient tiles take the radix-4 geometry.  Measured per ring
// (profiles/r02_experiments.txt for N = 2^15; profiles/r04_experiments.txt for 2^16 / 2^17: a grouped-digit hop at N = 2^17, level 12, takes
// 235 us with the N = 2^15 value and 205 with this one -- a limb there is 64 tiles, and the radix-8 tiles win from a handful of limbs on).
// option small_tile_wgs >= 0 overrides the table.
inline long small_tile_threshold(size_t N)
{
    const long o = (long)option(OPT_SMALL_TILE_WGS);
    if (o >= 0) return o;
    return N <= ((size_t)1 << 15) ? 5000 : N == ((size_t)1 << 16) ? kSmallTileWgsN16 : 400;
}
inline bool use_small_tiles(size_t N, long limbs) { return (long)(N >> kTileLog) * limbs < small_tile_threshold(N); }
// Round 5: a FOURTH geometry for throughput-bound forward COLS launches, LOGE = 4: 4096-coefficient tiles, 16 coefficients per thread, radix-16
// passes -- a 128-row phase (N = 2^15) is two passes (4 + 3 stages) with ONE LDS exchange instead of three (3 + 3 + 1) with two, and its row
// segments are 256 bytes.  Instantiated for the key switch's lift launches at k1 = 7 and 8 only (fused_ks.hip f_ks_lift_fcols; some other
// (K, direction) instantiations of a radix-16 tile take hipcc tens of minutes): in the 13-prime lowering f_ks_lift_fcols 177.7 -> 160.6 us
// per launch.  The generic COLS phase kernel at N = 2^17 gained nothing from it (hop at 31 primes 458-466 us either way) and keeps radix 8.
// option wide_tile_wgs: lift launches of at least this many 4096-coefficient tiles take it (-1: never).
inline bool use_wide_tiles(size_t N, long limbs)
{
    const long o = (long)option(OPT_WIDE_TILE_WGS);
    return o >= 0 && N >= 4096 && (long)(N >> 12) * limbs >= o;
}
// Round 5: ROWS-phase launches on rings whose twiddle tables outgrow the L2 walk their limbs prime by prime (ntt_kernels.hip ntt_phase_kernel,
// hybrid_fused.hip F1 / F9).  option rows_prime_major = log2 of the smallest such ring (16; 0 = never).
template <class Ctx>
inline bool rows_prime_major(const Ctx &c)
{
    const long o = (long)option(OPT_ROWS_PRIME_MAJOR);
    return o > 0 && c.logN >= o;
}
// in 512-coefficient workgroups (4 waves each): 256 = one wave on each of the 1024 SIMDs
inline long tiny_tile_threshold() { return (long)option(OPT_TINY_TILE_WGS); }
inline bool use_tiny_tiles(size_t N, long limbs) { return (long)(N >> TileGeo<1>::LOG) * limbs <= tiny_tile_threshold(); }

} // namespace dacapo
