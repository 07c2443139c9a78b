// One workgroup = one 2048-coefficient tile of one RNS limb; the negacyclic NTT of a limb of N = 2^logN
// coefficients is two launches ("phases") of this tile routine, so that a single limb already spreads over
// N/2048 workgroups (16 at the HEVM ring N = 2^15) and a key switch at small level still fills the chip:
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
// Inside a tile each thread keeps 8 coefficients in registers and runs up to three butterfly stages
// (radix-8) per pass; between passes the tile is transposed through LDS.  The LDS image is laid out for the
// READER: element j of thread t lives at j*(T+pad)+t, so every ds_read_b64 is lane-contiguous; the pad is
// chosen per exchange so the scattered ds_write_b64 of the writer is conflict-free as well
// (tools/lds_conflicts.py enumerates them).
#pragma once
#include "modarith.hpp"

namespace dacapo {

constexpr int kTileLog = 11;              // 2048 coefficients per workgroup
constexpr int kTileElems = 1 << kTileLog; // 16 KiB of u64
constexpr int kTileThreads = kTileElems / 8;

__host__ __device__ constexpr int pass_stages(int K, int p) { return (K - 3 * p) >= 3 ? 3 : (K - 3 * p); }
__host__ __device__ constexpr int num_passes(int K) { return (K + 2) / 3; }

// pad (in u64 elements) added to the register stride of the LDS image read by pass `p_reader`
template <int K, bool COLS>
__host__ __device__ constexpr int lds_pad(int p_reader, bool inverse);

// ---- per-pass index algebra (all compile-time foldable once loops are unrolled) ----------------------
template <int K>
struct PassMap {
    // thread sub-index s in [0, n/8), register j in [0,8)  ->  local coefficient index in [0, n)
    __device__ static __forceinline__ int idx_of(int p, int s, int j)
    {
        const int s0 = 3 * p, r = pass_stages(K, p);
        const int u = j >> r, kk = j & ((1 << r) - 1);
        const int vt = (s << (3 - r)) | u;
        const int lob = K - s0 - r;
        const int hi = vt >> lob, lo = vt & ((1 << lob) - 1);
        return (hi << (K - s0)) | (kk << lob) | lo;
    }
    // inverse of idx_of: local index -> (s, j) under pass p
    __device__ static __forceinline__ void sj_of(int p, int idx, int &s, int &j)
    {
        const int s0 = 3 * p, r = pass_stages(K, p);
        const int lob = K - s0 - r;
        const int lo = idx & ((1 << lob) - 1);
        const int kk = (idx >> lob) & ((1 << r) - 1);
        const int hi = idx >> (K - s0);
        const int vt = (hi << lob) | lo;
        s = vt >> (3 - r);
        j = ((vt & ((1 << (3 - r)) - 1)) << r) | kk;
    }
};

// Cooley-Tukey butterfly, values < 2^63 in and out:  (x, y) -> (x + w y, x - w y)
__device__ __forceinline__ void ct_bfly(u64 &x, u64 &y, u64 w, const DModulus &M)
{
    u64 xf = fold60(x, M.delta);            // < 2^60 + 2^31
    u64 t = mulmod_lazy(w, y, M.delta);     // < 2^62 (w < 2^60, y < 2^63: product < 2^123)
    x = xf + t;                             // < 2^63
    y = xf + (M.q << 2) - t;                // 4q > t : < 2^60 + 2^31 + 2^62
}
// Gentleman-Sande butterfly, values < 4q (< 2^62) in and out:  (x, y) -> (x + y, (x - y) w)
// (every value here is a canonical input, a fold60 result or a mulmod_lazy result, all < 4q = 2^62 - 4 delta)
__device__ __forceinline__ void gs_bfly(u64 &x, u64 &y, u64 w, const DModulus &M)
{
    u64 s = x + y;                          // < 2^63
    u64 d = x + (M.q << 2) - y;             // 4q > y ; < 2^62 + 2^62 = 2^63
    x = fold60(s, M.delta);
    y = mulmod_lazy(w, d, M.delta);         // w < 2^60, d < 2^63 ; result < 4q
}

// Ld: u64 operator()(int gidx)            coefficient gidx (0..N-1) of this limb, value < 2^62
// St: void operator()(int gidx, u64 v)    v canonical if CANON else lazy (< 2^63)
// PRELOADED: x[] already holds the first pass's coefficients (register j <-> idx_of(first pass, s, j)); ld unused.
// KEEP     : leave the result in x[] (canonical if CANON) instead of calling st.  The last pass of an inverse phase and
//            the first pass of a forward phase of the same shape use the same thread<->coefficient map, so an inverse
//            tile can hand its output to a forward tile in registers (the fused iNTT -> base change -> NTT kernels).
template <int K, bool COLS, bool INV, bool CANON, bool PRELOADED, bool KEEP, class Ld, class St>
__device__ __forceinline__ void ntt_tile_x(u64 (&x)[8], const DModulus M, const u64 *__restrict__ tw, int logN, int tile, Ld ld,
                                           St st, u64 *__restrict__ lds)
{
    constexpr int n = 1 << K, LOGB = kTileLog - K, B = 1 << LOGB, T = kTileThreads, SUBT = n / 8;
    constexpr int NP = num_passes(K);
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
    const int lane_id = tile * B + b;
    const u32 twroot = COLS ? 1u : ((1u << sh) + (u32)lane_id);
    auto gidx = [&](int idx) -> int { return COLS ? ((idx << sh) + lane_id) : ((lane_id << K) + idx); };

#pragma unroll
    for (int pp = 0; pp < NP; pp++) {
        const int p = INV ? (NP - 1 - pp) : pp;
        const int s0 = 3 * p, r = pass_stages(K, p);
        const bool first = (pp == 0), last = (pp == NP - 1);
        // ---- load
        if (first) {
            if (!PRELOADED) {
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = ld(gidx(PassMap<K>::idx_of(p, s, j)));
            }
        } else {
            const int stride = T + lds_pad<K, COLS>(p, INV);
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = lds[j * stride + t];
        }
        // ---- butterflies: 2^(3-r) independent radix-2^r networks per thread
#pragma unroll
        for (int u = 0; u < (1 << (3 - r)); u++) {
            const int vt = (s << (3 - r)) | u;
            const int hi = vt >> (K - s0 - r);
#pragma unroll
            for (int tt = 0; tt < r; tt++) {
                const int st_ = INV ? (r - 1 - tt) : tt; // stage within the pass
                const int gs = s0 + st_;                 // local stage index
                const int half = 1 << (r - 1 - st_);
#pragma unroll
                for (int g = 0; g < (1 << st_); g++) {
                    const u32 twi = (twroot << gs) + (u32)((hi << st_) | g);
                    u64 w = tw[twi];
#pragma unroll
                    for (int e = 0; e < half; e++) {
                        const int j0 = (u << r) | (g << (r - st_)) | e, j1 = j0 | half;
                        if (!INV)
                            ct_bfly(x[j0], x[j1], w, M);
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
                    for (int j = 0; j < 8; j++) x[j] = canon(x[j], M);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) st(gidx(PassMap<K>::idx_of(p, s, j)), CANON ? canon(x[j], M) : x[j]);
            }
        } else {
            const int pn = INV ? p - 1 : p + 1;
            const int stride = T + lds_pad<K, COLS>(pn, INV);
            if (!first) __syncthreads(); // everyone has finished reading the previous image
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int s2, j2;
                PassMap<K>::sj_of(pn, PassMap<K>::idx_of(p, s, j), s2, j2);
                const int t2 = COLS ? ((s2 << LOGB) | b) : (b * SUBT + s2);
                lds[j2 * stride + t2] = x[j];
            }
            __syncthreads();
        }
    }
}

template <int K, bool COLS, bool INV, bool CANON, class Ld, class St>
__device__ __forceinline__ void ntt_tile(const DModulus M, const u64 *__restrict__ tw, int logN, int tile, Ld ld, St st,
                                         u64 *__restrict__ lds)
{
    u64 x[8];
    ntt_tile_x<K, COLS, INV, CANON, false, false>(x, M, tw, logN, tile, ld, st, lds);
}

// polynomial index of register j of this thread in the first pass of a forward tile (= last pass of an inverse tile)
template <int K, bool COLS>
__device__ __forceinline__ int tile_first_pass_gidx(int logN, int tile, int j)
{
    constexpr int LOGB = kTileLog - K, B = 1 << LOGB, SUBT = (1 << K) / 8;
    const int t = threadIdx.x;
    const int b = COLS ? (t & (B - 1)) : (t / SUBT), s = COLS ? (t >> LOGB) : (t & (SUBT - 1));
    const int idx = PassMap<K>::idx_of(0, s, j);
    const int lane_id = tile * B + b;
    return COLS ? ((idx << (logN - K)) + lane_id) : ((lane_id << K) + idx);
}

// Pads found with tools/lds_conflicts.py (0 = already conflict-free).  Max pad bounds the LDS allocation.
constexpr int kMaxLdsPad = 8;
constexpr int kTileLdsElems = 8 * (kTileThreads + kMaxLdsPad);

template <int K, bool COLS>
__host__ __device__ constexpr int lds_pad(int p_reader, bool inverse)
{
    (void)p_reader;
    (void)inverse;
    return 4;
}

} // namespace dacapo
