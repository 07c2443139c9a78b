// EXTENSION (see hybrid_ks.hip): the FUSED launch sequence of grouped-digit hybrid key switching -- round 4.  Same arithmetic as
// hybrid_ks.hip's nine-kernel-plus-four-transforms sequence (which stays selectable, option hyb_fuse = 0, as a second implementation) and as
// oracle/ckks_oracle.c orc_keyswitch_hybrid: every value written is the canonical residue of the same integer, so limbs are bit-identical.
// What it serves in the reference: HEAAN_HEVM.cpp:300-303 (rotate), :386-399 (bootstrap).
//
//   round 3                                  | here
//   prepare: c0' = galois(c0), digits = c1   | --   (F1 reads c1 in place; F9 gathers c0 through the Galois map in its epilogue)
//   iNTT digits            (2 launches)      | F1   inverse ROWS phase, out of place, once per decomposition (first item of a slot)
//                                            | F2   inverse COLS phase whose last stage multiplies by N^-1 qhat_inv_i (Context::hyb_upmods):
//                                            |      the conversion's per-input constant costs nothing
//   mod-up                 (1)               | F3   forward COLS phase of every raised limb with the conversion as its LOADER: sum_t y_t w[t][e]
//   NTT ext                (2)               |      in three 64-bit columns of 30-bit limb products (4 mads per term, no carries)
//                                            | F4   forward ROWS phase
//   mac                    (1)               | F5   inner products with the key (hyb_mac_kernel; a rotation's base term P galois(c0) joins the accumulator here)
//   iNTT acc_P             (2)               | F6   inverse ROWS;  F7  inverse COLS with N^-1 phat_inv_j folded (Context::d_hyb_dnmods)
//   mod-down               (1)               | F8   forward COLS phase of the 2 l correction limbs with the conversion as its loader
//   NTT tmp                (2)               | F9   forward ROWS phase with (acc - t) P^-1 + base as its store epilogue
//   final                  (1)               |
//   13 launches; a raised limb crosses HBM 6 times (written by mod-up, read + written by either NTT phase, read by mac)
//                                            | 9 launches; 4 crossings (F3 w, F4 r + w, F5 r); prepare's 4 l, mod-down's and final's limbs gone
// option hyb_fuse = 2 keeps the two conversions as separate matrix-core launches (hyb_conv_mfma_kernel on pre-scaled inputs) inside this
// sequence: F3 / F8 become conversion + two-phase transform.  Which of 1 / 2 is the default is decided by measurement.
#include "plan.hpp"
#include "tile_dispatch.hpp"

namespace dacapo {

__device__ __forceinline__ u32 hf_galois_idx(u32 k, u32 elt, int logN)
{ // GaloisTool::apply_galois_ntt index map (poly_kernels.hip)
    const u32 r = (__brev(k) >> (32 - logN)) * 2u + 1u;
    const u32 idx = ((elt * r) >> 1) & ((1u << logN) - 1u);
    return __brev(idx) >> (32 - logN);
}

// sum of up to 8 products y w with y, w < 2^60, kept as three 64-bit columns of 30-bit limb products: y = y0 + y1 2^30, w = w0 + w1 2^30,
// every product < 2^60, so c0 < 8 2^60, c1 < 16 2^60 <= 2^64 - 16 2^31, c2 < 8 2^60 never overflow and a term is 4 v_mad_u64_u32 with no
// carry handling (a 128-bit accumulator: 4 mads + 6 adds with carries).  w is wave-uniform in every use (a conversion constant): scalar operands.
struct Acc3 {
    u64 c0, c1, c2;
    __device__ __forceinline__ void clear() { c0 = c1 = c2 = 0; }
    __device__ __forceinline__ void mac(u64 y, u32 w0, u32 w1)
    {
        const u32 y0 = lo32(y) & 0x3FFFFFFFu, y1 = (u32)(y >> 30);
        c0 = mad32(y0, w0, c0);
        c1 = mad32(y0, w1, c1);
        c1 = mad32(y1, w0, c1);
        c2 = mad32(y1, w1, c2);
    }
    // T = c0 + c1 2^30 + c2 2^60 < 2^124 -> canonical residue
    __device__ __forceinline__ u64 reduce(const DModulus &M) const
    {
        u64 lo = c0 + (c1 << 30);
        u64 hi = (c1 >> 34) + (lo < c0 ? 1u : 0u);
        const u64 l2 = lo + (c2 << 60);
        hi += (c2 >> 4) + (l2 < lo ? 1u : 0u);
        return reduce128_any(hi, l2, M);
    }
};
// inputs of a conversion whose loads are in flight together: 4 x 8 coefficients (64 VGPRs) for the radix-8 tiles, 8 x 4 / 8 x 2 below
template <int LOGE>
constexpr int kConvChunk = LOGE == 3 ? 4 : 8;
__device__ __forceinline__ u32 w_lo30(u64 w) { return (u32)w & 0x3FFFFFFFu; }
__device__ __forceinline__ u32 w_hi30(u64 w) { return (u32)(w >> 30); }

// true for the first item of the batch that names this item's decomposition slot (wave-uniform; every wave of the workgroup agrees)
__device__ __forceinline__ bool hf_first_of_slot(const KsItem *__restrict__ items, int b)
{
    const u32 mine = items[b].slot;
    const int lane = threadIdx.x & 63;
    for (int base = 0; base < b; base += 64) {
        const int j = base + lane;
        const bool hit = j < b && items[j].slot == mine;
        if (__ballot(hit)) return false;
    }
    return true;
}

// ---- F1: inverse ROWS phase of the decomposition's source limbs, out of place.  z = b * ell + i.
// MODE 0: rotation items -- src.c1 as it is (the digits are taken BEFORE the automorphism), once per slot; MODE 1: strided source
// (target [B][ell][N]: ct x ct's c2, or one key switch by value)
template <int K, int LOGE, int MODE>
__global__ __launch_bounds__(kTileThreads) void hybf_irows_kernel(const KsItem *__restrict__ items, KsItem single, const u64 *__restrict__ target,
                                                                   u64 *__restrict__ digits, int ell, int use_slots,
                                                                   const DModulus *__restrict__ mods, const u64 *__restrict__ itw, int logN,
                                                                   int prime_major)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    // (prime_major: the B limbs of one prime adjacent in launch order -- their twiddle tiles come out of L2, ntt_kernels.hip)
    const int B = gridDim.y / ell;
    const int i = prime_major ? blockIdx.y / B : blockIdx.y % ell, b = prime_major ? blockIdx.y % B : blockIdx.y / ell, z = b * ell + i;
    const size_t N = (size_t)1 << logN;
    const u64 *in;
    size_t slot = (size_t)b;
    if (MODE == 0) {
        if (items && use_slots) {
            if (!hf_first_of_slot(items, b)) return;
            slot = items[b].slot;
        }
        const KsItem it = items ? items[b] : single;
        in = it.src.limb(1, i, N);
    } else
        in = target + (size_t)z * N;
    u64 *out = digits + (slot * ell + i) * N;
    ntt_tile<K, LOGE, false, true, false>(
        mods[i], itw + ((size_t)i << logN), logN, blockIdx.x, [=](int g) { return in[g]; }, [=](int g, u64 v) { out[g] = v; }, lds);
}

// ---- F3: mod-up as the loader of the raised limbs' first forward phase.  y = blockIdx.y = u * E + r (decomposition u, raised limb r in
// the order of Context::hyb_pidx: digit by digit, the other moduli ascending).  digits [U][ell][N]: y_t = [x_t qhat_t^-1]_{q_t} (F2).
template <int K, int LOGE>
__global__ __launch_bounds__(kTileThreads) void hybf_modup_fcols_kernel(const u64 *__restrict__ digits, u64 *__restrict__ ext, int ell, int ksp,
                                                                         int alpha, int L, int E, const DModulus *__restrict__ mods,
                                                                         const u64 *__restrict__ up, const u64 *__restrict__ tw, int logN)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    constexpr int EC = 1 << LOGE;
    const size_t N = (size_t)1 << logN;
    const int r = blockIdx.y % E, u = blockIdx.y / E, M = ell + ksp, full = M - alpha, G = (ell + alpha - 1) / alpha;
    int g = r / full;
    g = g < G ? g : G - 1; // (the last digit may be partial: it has more than `full` other moduli)
    const int idx = r - g * full, lo = g * alpha, hi = min(lo + alpha, ell), a = hi - lo;
    const int mi = idx < lo ? idx : idx + a, pm = mi < ell ? mi : L + (mi - ell);
    const DModulus Mo = mods[pm];
    const u64 *w = up + ell + (size_t)lo * M + mi; // w[t * M] = (Q_g / q_{lo+t}) mod m
    const u64 *src = digits + ((size_t)u * ell + lo) * N;
    int gi[EC];
#pragma unroll
    for (int j = 0; j < EC; j++) gi[j] = tile_gidx<K, LOGE, true>(0, logN, blockIdx.x, j);
    Acc3 acc[EC];
#pragma unroll
    for (int j = 0; j < EC; j++) acc[j].clear();
    u64 x[EC];
#pragma unroll
    for (int j = 0; j < EC; j++) x[j] = 0;
    // the loads of kConvChunk inputs are issued together (one memory latency per chunk, not per input: the first version waited seven times)
    for (int t0 = 0; t0 < a; t0 += kConvChunk<LOGE>) {
        u64 y[kConvChunk<LOGE>][EC];
#pragma unroll
        for (int tt = 0; tt < kConvChunk<LOGE>; tt++) {
            if (t0 + tt < a) {
#pragma unroll
                for (int j = 0; j < EC; j++) y[tt][j] = src[(size_t)(t0 + tt) * N + gi[j]];
            }
        }
#pragma unroll
        for (int tt = 0; tt < kConvChunk<LOGE>; tt++) {
            if (t0 + tt < a) {
                const u64 c = w[(size_t)(t0 + tt) * M];
                const u32 c0 = w_lo30(c), c1 = w_hi30(c);
#pragma unroll
                for (int j = 0; j < EC; j++) acc[j].mac(y[tt][j], c0, c1);
            }
        }
        if (((t0 + kConvChunk<LOGE>) & 7) == 0 && t0 + kConvChunk<LOGE> < a) { // more than 8 inputs (alpha up to 16): bank the columns
#pragma unroll
            for (int j = 0; j < EC; j++) x[j] = addmod(x[j], acc[j].reduce(Mo), Mo.q), acc[j].clear();
        }
    }
#pragma unroll
    for (int j = 0; j < EC; j++) x[j] = addmod(x[j], acc[j].reduce(Mo), Mo.q);
    u64 *out = ext + (size_t)blockIdx.y * N;
    auto nold = [](int) -> u64 { return 0; };
    ntt_tile_x<K, LOGE, true, false, false, true, false>(x, Mo, tw + ((size_t)pm << logN), logN, blockIdx.x, nold, [=](int gg, u64 v) { out[gg] = v; }, lds);
}

// ---- F7: second inverse phase of the special-prime accumulators, in place.  z = poly * ksp + j.  The constants passed in `dnmods` carry
// N^-1 phat_j^-1, and the store adds floor(P/2) phat_j^-1 mod p_j: what leaves is [(r_j + floor(P/2)) phat_j^-1]_{p_j}, the mod-down's input,
// computed once per coefficient instead of once per (coefficient, output modulus)
template <int K, int LOGE>
__global__ __launch_bounds__(kTileThreads) void hybf_icols_special_kernel(u64 *__restrict__ accp, int ksp, int L, const DModulus *__restrict__ dnmods,
                                                                           const u64 *__restrict__ hp, const u64 *__restrict__ itw, int logN)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    const int j = blockIdx.y % ksp, p = L + j;
    const DModulus M = dnmods[p];
    const u64 h = hp[j];
    u64 *a = accp + ((size_t)blockIdx.y << logN);
    ntt_tile<K, LOGE, true, true, true>(
        M, itw + ((size_t)p << logN), logN, blockIdx.x, [=](int g) { return a[g]; }, [=](int g, u64 v) { a[g] = addmod(v, h, M.q); }, lds);
}

// ---- F8: mod-down as the loader of the correction limbs' first forward phase.  y = z * ell + i (polynomial z = 2 b + c, data prime i).
// accp [2B][ksp][N]: z_j = [(r_j + floor(P/2)) phat_j^-1]_{p_j} (F7); t_i = sum_j z_j (P / p_j) - floor(P/2)  mod q_i
template <int K, int LOGE>
__global__ __launch_bounds__(kTileThreads) void hybf_moddown_fcols_kernel(const u64 *__restrict__ accp, u64 *__restrict__ tmp, int ell, int ksp,
                                                                           int L, const DModulus *__restrict__ mods, const u64 *__restrict__ dn,
                                                                           const u64 *__restrict__ tw, int logN)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    constexpr int EC = 1 << LOGE;
    const size_t N = (size_t)1 << logN;
    const int i = blockIdx.y % ell, z = blockIdx.y / ell;
    const DModulus Mo = mods[i];
    const u64 half_q = dn[2 * ksp + i];
    const u64 *w = dn + 2 * ksp + 2 * L + i; // w[j * L] = (P / p_j) mod q_i
    const u64 *src = accp + (size_t)z * ksp * N;
    int gi[EC];
#pragma unroll
    for (int j = 0; j < EC; j++) gi[j] = tile_gidx<K, LOGE, true>(0, logN, blockIdx.x, j);
    Acc3 acc[EC];
    u64 x[EC];
#pragma unroll
    for (int j = 0; j < EC; j++) acc[j].clear(), x[j] = 0;
    for (int t0 = 0; t0 < ksp; t0 += kConvChunk<LOGE>) {
        u64 y[kConvChunk<LOGE>][EC];
#pragma unroll
        for (int tt = 0; tt < kConvChunk<LOGE>; tt++) {
            if (t0 + tt < ksp) {
#pragma unroll
                for (int j = 0; j < EC; j++) y[tt][j] = src[(size_t)(t0 + tt) * N + gi[j]];
            }
        }
#pragma unroll
        for (int tt = 0; tt < kConvChunk<LOGE>; tt++) {
            if (t0 + tt < ksp) {
                const u64 c = w[(size_t)(t0 + tt) * L];
                const u32 c0 = w_lo30(c), c1 = w_hi30(c);
#pragma unroll
                for (int j = 0; j < EC; j++) acc[j].mac(y[tt][j], c0, c1);
            }
        }
        if (((t0 + kConvChunk<LOGE>) & 7) == 0 && t0 + kConvChunk<LOGE> < ksp) {
#pragma unroll
            for (int j = 0; j < EC; j++) x[j] = addmod(x[j], acc[j].reduce(Mo), Mo.q), acc[j].clear();
        }
    }
#pragma unroll
    for (int j = 0; j < EC; j++) x[j] = submod(addmod(x[j], acc[j].reduce(Mo), Mo.q), half_q, Mo.q);
    u64 *out = tmp + (size_t)blockIdx.y * N;
    auto nold = [](int) -> u64 { return 0; };
    ntt_tile_x<K, LOGE, true, false, false, true, false>(x, Mo, tw + ((size_t)i << logN), logN, blockIdx.x, nold, [=](int gg, u64 v) { out[gg] = v; }, lds);
}

struct HybOut { // one key switch by value: out = (base0, base1) + KS(target)
    CtView out;
    const u64 *base0 = nullptr, *base1 = nullptr;
};

// ---- F9: last forward phase of the correction limbs with dst.c = base + (acc_c - t_c) P^-1 as its epilogue.  y = z * ell + i.
// MODE 0 rotation items: no base here -- F5 has added P galois(c0) to the accumulator (hyb_mac_kernel, fold_base); 1 ct x ct items (base: the
// tensor product's c0 / c1 already in dst), 2 one key switch by value.
// After its last pass a ROWS tile leaves every thread with E CONSECUTIVE coefficients (idx = (s << LOGE) | j), so the epilogue's three streams
// (accumulator, base, destination) are 16-byte accesses, 64 contiguous bytes per thread.  (The first version of this kernel called a
// per-coefficient store functor with a Galois gather inside: 8-byte accesses at a 64-byte lane stride, and the gather's 64 lanes in 64 different
// cache lines -- 1 460 us per launch in the config-4 run against 570 for the separate transform + element-wise kernel it replaced.)
template <int K, int LOGE, int MODE>
__global__ __launch_bounds__(kTileThreads) void hybf_frows_final_kernel(const u64 *__restrict__ tmp, const u64 *__restrict__ accq,
                                                                         const void *__restrict__ items, KsItem rot_single, HybOut single, int ell,
                                                                         int ksp, int L, const DModulus *__restrict__ mods,
                                                                         const u64 *__restrict__ dn, const u64 *__restrict__ tw, int logN,
                                                                         int prime_major)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    constexpr int EC = 1 << LOGE, NP = num_passes<LOGE>(K);
    const size_t N = (size_t)1 << logN;
    // (prime_major: the 2 B limbs of one prime adjacent in launch order -- their twiddle tiles come out of L2, ntt_kernels.hip)
    const int polys = gridDim.y / ell;
    const int i = prime_major ? blockIdx.y / polys : blockIdx.y % ell, z = prime_major ? blockIdx.y % polys : blockIdx.y / ell, b = z >> 1, c = z & 1;
    const DModulus M = mods[i];
    const u64 pinv = dn[2 * ksp + L + i];
    const u64 *in = tmp + ((size_t)z * ell + i) * N, *ac = accq + ((size_t)z * ell + i) * N;
    u64 *dst;
    const u64 *base = nullptr;
    if (MODE == 0) {
        const KsItem it = items ? static_cast<const KsItem *>(items)[b] : rot_single;
        dst = it.dst.limb(c, i, N);
    } else if (MODE == 1) {
        const MulItem it = static_cast<const MulItem *>(items)[b];
        dst = it.dst.limb(c, i, N), base = dst;
    } else {
        dst = single.out.limb(c, i, N);
        const u64 *bp = c == 0 ? single.base0 : single.base1;
        base = bp ? bp + (size_t)i * N : nullptr;
    }
    u64 x[EC];
    auto nost = [](int, u64) {};
    ntt_tile_x<K, LOGE, false, false, true, false, true>(x, M, tw + ((size_t)i << logN), logN, blockIdx.x, [=](int g) { return in[g]; }, nost, lds);
    const int g0 = tile_gidx<K, LOGE, false>(NP - 1, logN, blockIdx.x, 0); // register j holds coefficient g0 + j
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < EC; j += 2) {
        const u64x2 a = *reinterpret_cast<const u64x2 *>(ac + g0 + j);
        u64x2 r;
        r.x = mulmod(submod(a.x, x[j], M.q), pinv, M), r.y = mulmod(submod(a.y, x[j + 1], M.q), pinv, M);
        if (base) {
            const u64x2 o = *reinterpret_cast<const u64x2 *>(base + g0 + j);
            r.x = addmod(r.x, o.x, M.q), r.y = addmod(r.y, o.y, M.q);
        }
        *reinterpret_cast<u64x2 *>(dst + g0 + j) = r;
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------------------
template <int MODE>
static void f1_irows(const Context &c, const KsItem *items, KsItem single, const u64 *target, u64 *digits, int B, int ell, int use_slots, hipStream_t s)
{
    DC_GEO_SWITCH(c.k2, B * ell, DC_LAUNCH((hybf_irows_kernel<KK, LE, MODE>), grid, dim3(kTileThreads), 0, s, items, single, target, digits, ell,
                                                    use_slots, c.d_mods, c.d_itw, c.logN, (int)(B > 1 && rows_prime_major(c))));
}
static void f3_modup_fcols(const Context &c, const u64 *digits, u64 *ext, int U, int ell, hipStream_t s)
{
    const int E = c.hyb_ext(ell);
    DC_GEO_SWITCH(c.k1, U * E, DC_LAUNCH((hybf_modup_fcols_kernel<KK, LE>), grid, dim3(kTileThreads), 0, s, digits, ext, ell, c.ksp, c.alpha,
                                                  c.max_level(), E, c.d_mods, c.hyb_up(ell), c.d_tw, c.logN));
}
static void f8_moddown_fcols(const Context &c, const u64 *accp, u64 *tmp, int polys, int ell, hipStream_t s)
{
    DC_GEO_SWITCH(c.k1, polys * ell, DC_LAUNCH((hybf_moddown_fcols_kernel<KK, LE>), grid, dim3(kTileThreads), 0, s, accp, tmp, ell, c.ksp,
                                                        c.max_level(), c.d_mods, c.d_hyb_dn, c.d_tw, c.logN));
}
static void f7_icols_special(const Context &c, u64 *accp, int polys, hipStream_t s)
{
    DC_GEO_SWITCH(c.k1, polys * c.ksp, DC_LAUNCH((hybf_icols_special_kernel<KK, LE>), grid, dim3(kTileThreads), 0, s, accp, c.ksp, c.max_level(),
                                                          c.d_hyb_dnmods, c.d_hyb_hp, c.d_itw, c.logN));
}
template <int MODE>
static void f9_frows_final(const Context &c, const u64 *tmp, const u64 *accq, const void *items, KsItem rot_single, HybOut single, int polys, int ell,
                           hipStream_t s)
{
    DC_GEO_SWITCH(c.k2, polys * ell, DC_LAUNCH((hybf_frows_final_kernel<KK, LE, MODE>), grid, dim3(kTileThreads), 0, s, tmp, accq, items,
                                                        rot_single, single, ell, c.ksp, c.max_level(), c.d_mods, c.d_hyb_dn, c.d_tw, c.logN,
                                                        (int)rows_prime_major(c)));
}

// hybrid_ks.hip
void hyb_launch_mac(Context &c, int mode, const BatchWs &w, const void *items, KsItem rot_single, const u64 *key, int B, int use_slots, int ell,
                    hipStream_t s, bool fold_base);
void hyb_launch_conv(Context &c, bool down, bool prescaled, const u64 *in, u64 *out, int count, int ell, hipStream_t s);
void hyb_launch_mac_groups(Context &c, const BatchWs &w, const KsItem *items, const KsItem *groups, int G, int use_slots, int ell, hipStream_t s);

// F2 ... F5: everything between F1 and the accumulators.  digits [U][ell][N] hold the inverse ROWS phase's output.
template <int MODE>
static void hybf_front(Context &c, const BatchWs &w, const void *items, KsItem rot_single, const u64 *key, int B, int U, int use_slots, int ell,
                       hipStream_t s, bool with_mac = true)
{
    const size_t N = c.N;
    const int E = c.hyb_ext(ell);
    const bool separate_conv = option(OPT_HYB_FUSE) == 2 && c.hyb_mfma && N >= 512 && ell >= 4;
    launch_ntt_cols_inv(c, w.digits, (long)N, U * ell, nullptr, 0, ell, s, c.hyb_upmods(ell));                    // F2
    if (separate_conv) {
        hyb_launch_conv(c, false, true, w.digits, w.ext, U, ell, s);
        launch_ntt(c, false, w.ext, (long)N, U * E, c.hyb_pidx(ell), 0, E, s);
    } else {
        f3_modup_fcols(c, w.digits, w.ext, U, ell, s);                                                             // F3
        launch_ntt_rows_fwd(c, w.ext, (long)N, U * E, c.hyb_pidx(ell), 0, E, s);                                   // F4
    }
    if (with_mac) hyb_launch_mac(c, MODE, w, items, rot_single, key, B, use_slots, ell, s, MODE == 0);             // F5
}
// F6 ... F9: the division by P of `B` accumulator pairs accq [2B][ell][N], accp [2B][ksp][N]; items[b] names where pair b goes
template <int MODE>
static void hybf_back(Context &c, const BatchWs &w, const u64 *accq, u64 *accp, const void *items, KsItem rot_single, HybOut single, int B, int ell,
                      hipStream_t s)
{
    const size_t N = c.N;
    const int ksp = c.ksp, L = c.max_level();
    const bool separate_conv = option(OPT_HYB_FUSE) == 2 && c.hyb_mfma && N >= 512 && ell >= 4;
    launch_ntt_rows_inv(c, accp, (long)N, 2 * B * ksp, nullptr, L, ksp, s);                                        // F6
    f7_icols_special(c, accp, 2 * B, s);                                                                           // F7
    if (separate_conv) {
        hyb_launch_conv(c, true, true, accp, w.tmp, 2 * B, ell, s);
        launch_ntt_cols_fwd(c, w.tmp, (long)N, 2 * B * ell, nullptr, 0, ell, s);
    } else
        f8_moddown_fcols(c, accp, w.tmp, 2 * B, ell, s);                                                           // F8
    f9_frows_final<MODE>(c, w.tmp, accq, items, rot_single, single, 2 * B, ell, s);                                // F9
}
template <int MODE>
static void hybf_core(Context &c, const BatchWs &w, const void *items, KsItem rot_single, HybOut single, const u64 *key, int B, int U, int use_slots,
                      int ell, hipStream_t s)
{
    hybf_front<MODE>(c, w, items, rot_single, key, B, U, use_slots, ell, s);
    hybf_back<MODE>(c, w, w.acc, w.acc + (size_t)B * 2 * ell * c.N, items, rot_single, single, B, ell, s);
}

// ---- lazy sums (option hyb_lazy_sum; oracle/ckks_oracle.c orc_rotate_acc_hybrid / orc_moddown_hybrid) --------------------------------------
// The rotations of one step whose results are only ever ADDED together (the giant steps of a BSGS matrix-vector product) share ONE division
// by P: F1 ... F5 run for every item as usual, this kernel adds the accumulators of a group's items (canonical residues, every one of the
// l + ksp limbs, both polynomials; each item's base term P galois(c0) is already inside its accumulator), and F6 ... F9 run once per GROUP.
// groups[g]: dst = where the sum goes, elt = the group's first item, slot = its item count (the items of a group are adjacent).
// grid = (N / 512, l + ksp, 2 G).  Reads 2 (l + ksp) limbs per item once: 16 us per item at l = 31 against the ~140 us of the F6 ... F9 it saves.
__global__ __launch_bounds__(256) void hybf_group_sum_kernel(const u64 *__restrict__ accq, const u64 *__restrict__ accp, u64 *__restrict__ gq,
                                                             u64 *__restrict__ gp, const KsItem *__restrict__ items, const KsItem *__restrict__ groups,
                                                             int ell, int ksp, int L, size_t N, const DModulus *__restrict__ mods)
{
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    const int mi = blockIdx.y, g = blockIdx.z >> 1, c = blockIdx.z & 1;
    const u32 first = groups[g].elt, count = groups[g].slot;
    const bool special = mi >= ell;
    const int limbs = special ? ksp : ell, row = special ? mi - ell : mi;
    const DModulus M = mods[special ? L + row : row];
    const u64 q = M.q;
    const size_t k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2, stride = (size_t)2 * limbs * N;
    const u64 *in = (special ? accp : accq) + (((size_t)first * 2 + c) * limbs + row) * N + k;
    // an item's accumulator, times its plaintext when it has one (double hoisting: hyb_mac_group_kernel's rule, hybrid_ks.hip)
    auto term = [&](u32 t) {
        u64x2 v = *reinterpret_cast<const u64x2 *>(in + (size_t)t * stride);
        const KsItem &it = items[first + t];
        if (it.plain) {
            const u64x2 w = *reinterpret_cast<const u64x2 *>((special ? it.plain_sp : it.plain) + (size_t)row * N + k);
            v.x = mulmod(v.x, w.x, M), v.y = mulmod(v.y, w.y, M);
        }
        return v;
    };
    u64x2 a = term(0);
    u32 t = 1;
    for (; t + 1 < count; t += 2) { // two items' loads in flight
        const u64x2 v0 = term(t), v1 = term(t + 1);
        a.x = addmod(addmod(a.x, v0.x, q), v1.x, q), a.y = addmod(addmod(a.y, v0.y, q), v1.y, q);
    }
    if (t < count) {
        const u64x2 v = term(t);
        a.x = addmod(a.x, v.x, q), a.y = addmod(a.y, v.y, q);
    }
    *reinterpret_cast<u64x2 *>((special ? gp : gq) + (((size_t)g * 2 + c) * limbs + row) * N + k) = a;
}

// B rotation items in G groups -> G sums.  Default: the inner-product kernel itself walks a group's items (hyb_mac_group_kernel: no item's
// accumulator is ever stored).  option hyb_lazy_sum = 2 -- a second implementation of the same sums, kept for the parity tests: every item's
// accumulators as for an ordinary step, added by hybf_group_sum_kernel into w.ext (free once F5 has read the raised limbs; 2 G (l + ksp) <= B E
// limbs: every group has at least two items).
void hybf_rotate_sum(Context &c, const BatchWs &w, const KsItem *d_items, int B, const KsItem *d_groups, int G, int ell, hipStream_t s, int unique)
{
    const size_t N = c.N;
    const int use_slots = unique > 0 ? 1 : 0, U = use_slots ? unique : B, ksp = c.ksp;
    const bool separate_sum = option(OPT_HYB_LAZY_SUM) == 2;
    if ((size_t)2 * G * (ell + ksp) > (size_t)B * c.hyb_ext(ell) || N < 512 || 2 * G > B) {
        fprintf(stderr, "[dacapo_amd] lazy sum: %d groups of %d items do not fit the step's scratch (internal error)\n", G, B);
        abort();
    }
    f1_irows<0>(c, d_items, KsItem{}, nullptr, w.digits, B, ell, use_slots, s);
    hybf_front<0>(c, w, d_items, KsItem{}, nullptr, B, U, use_slots, ell, s, separate_sum);
    if (!separate_sum) {
        hyb_launch_mac_groups(c, w, d_items, d_groups, G, use_slots, ell, s);
        hybf_back<0>(c, w, w.acc, w.acc + (size_t)G * 2 * ell * N, d_groups, KsItem{}, HybOut{}, G, ell, s);
        return;
    }
    u64 *gq = w.ext, *gp = w.ext + (size_t)2 * G * ell * N;
    DC_LAUNCH(hybf_group_sum_kernel, dim3((unsigned)(N / 512), (unsigned)(ell + ksp), (unsigned)(2 * G)), dim3(256), 0, s, w.acc,
              w.acc + (size_t)B * 2 * ell * N, gq, gp, d_items, d_groups, ell, ksp, c.max_level(), N, c.d_mods);
    hybf_back<0>(c, w, gq, gp, d_groups, KsItem{}, HybOut{}, G, ell, s);
}

void hybf_rotate_hops(Context &c, const BatchWs &w, const KsItem *d_items, int B, int ell, hipStream_t s, int unique)
{
    const int use_slots = unique > 0 ? 1 : 0, U = use_slots ? unique : B;
    f1_irows<0>(c, d_items, KsItem{}, nullptr, w.digits, B, ell, use_slots, s);
    hybf_core<0>(c, w, d_items, KsItem{}, HybOut{}, nullptr, B, U, use_slots, ell, s);
}

void hybf_rotate_hop_single(Context &c, const Workspace &w, CtView dst, CtView src, u32 galois_elt, const u64 *galois_key, int ell, hipStream_t s)
{
    const KsItem it{ src, dst, galois_key, galois_elt, 0 };
    BatchWs bw{ nullptr, w.ks_digits, w.ks_ext, w.ks_acc, w.ks_tmp };
    f1_irows<0>(c, nullptr, it, nullptr, w.ks_digits, 1, ell, 0, s);
    hybf_core<0>(c, bw, nullptr, it, HybOut{}, nullptr, 1, 1, 0, ell, s);
}

// target [B][ell][N] = the tensor products' c2 (NTT form), dst.c0 / dst.c1 hold their c0 / c1 (hyb_prepare_mul_kernel has run)
void hybf_mul_relin_tail(Context &c, const BatchWs &w, const MulItem *d_items, const u64 *relin_key, int B, int ell, hipStream_t s)
{
    f1_irows<1>(c, nullptr, KsItem{}, w.target, w.digits, B, ell, 0, s);
    hybf_core<1>(c, w, d_items, KsItem{}, HybOut{}, relin_key, B, B, 0, ell, s);
}

void hybf_keyswitch(Context &c, const Workspace &w, CtView out, const u64 *base0, const u64 *base1, const u64 *target, const u64 *key, int ell,
                    hipStream_t s)
{
    BatchWs bw{ const_cast<u64 *>(target), w.ks_digits, w.ks_ext, w.ks_acc, w.ks_tmp };
    f1_irows<1>(c, nullptr, KsItem{}, target, w.ks_digits, 1, ell, 0, s);
    hybf_core<2>(c, bw, nullptr, KsItem{}, HybOut{ out, base0, base1 }, key, 1, 1, 0, ell, s);
}

} // namespace dacapo
