// Fused phase kernels of the batched key-switch / rescale pipelines (plan.hpp, batch_ops.hip).
//
// A key switch at level l is  iNTT(l limbs) -> lift each digit to the l other moduli -> NTT(l*l limbs) -> inner
// products with the key -> mod-down (iNTT of the special-prime limb, lift, NTT, scale) [SEAL-upstream
// Evaluator::switch_key_inplace, reached from SEAL_HEVM.cpp:273,316].  Launched phase by phase that is 13 dependent
// launches; at the HEVM working levels (2-5 primes) each is microseconds of work, so the launch chain IS the latency.
// The tile routine lets an inverse COLS phase hand its canonical output to a forward COLS phase in registers, so:
//   L1 irows   : inverse ROWS phase, reading the operand in place (Galois gather fused into the loader)
//   L2 icols+lift+fcols : per (digit j, other modulus e): finish the iNTT of digit j, reduce into q_e, first NTT phase
//   L3 frows   : second NTT phase over the l*l lifted digits
//   L4 mac     : inner products with the key (the j == m term reads the operand in place)
//   L5 irows   : special-prime limb of both accumulators
//   L6 icols+round+fcols : finish its iNTT, add floor(P/2), reduce into q_i, subtract, first NTT phase
//   L7 frows+final : second NTT phase, (acc - t) * P^-1 + base, written straight into the destination
// 7 launches instead of 13; rescale is L5'-L7' = 3 instead of 7.  (Round 3: throughput-bound launches take MERGE instantiations of L2 / L6
// / L3-L5 that do not repeat work per target modulus / per accumulator; see the kernels.)  L2/L6 recompute the inverse COLS phase once per
// target modulus (it is 1/(l+1) of that kernel's work) to keep every workgroup at two phases.
#include "ntt_tile.hpp"
#include "plan.hpp"
#include "tile_dispatch.hpp"

namespace dacapo {

__device__ __forceinline__ u32 galois_idx(u32 k, u32 elt, int logN)
{
    const u32 r = (__brev(k) >> (32 - logN)) * 2u + 1u;
    const u32 idx = ((elt * r) >> 1) & ((1u << logN) - 1u);
    return __brev(idx) >> (32 - logN);
}

// E consecutive coefficients k0 .. k0 + E - 1 (k0 a multiple of E >= 2) of galois(p): the index map sends an aligned pair of outputs to an
// aligned pair of inputs, possibly swapped (brev(k + 1) = brev(k) + N/2, elt odd: the source index moves by N/2 before its own bit
// reversal, i.e. its lowest bit flips), so a pair is ONE 16-byte load
template <int E>
__device__ __forceinline__ void galois_gather(u64 (&x)[E], const u64 *__restrict__ p, u32 k0, u32 elt, int logN)
{
    static_assert(E >= 2 && E % 2 == 0, "pairs");
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int e = 0; e < E; e += 2) {
        const u32 gi = galois_idx(k0 + (u32)e, elt, logN);
        const u64x2 v = *reinterpret_cast<const u64x2 *>(p + (gi & ~1u));
        x[e] = (gi & 1u) ? v.y : v.x, x[e + 1] = (gi & 1u) ? v.x : v.y;
    }
}

// ---- operand sources of the first inverse phase ----------------------------------------------------------------------
struct SrcStrided { // limb z at base + z*stride, modulo prime prime_base + z % period
    const u64 *base;
    long stride;
    int prime_base, period;
    __device__ int prime(int z) const { return prime_base + z % period; }
    __device__ u64 load(int z, int g, int) const { return base[(long)z * stride + g]; }
};
// Operand of a rescale item (plan.hpp RsItem) for the E coefficients a thread owns: elementwise producers (n-ary sum,
// + plaintext, * plaintext) that nothing else reads are evaluated here instead of being materialised.  Modular
// arithmetic is exact, so the values equal what the separate ops would have stored.  Every stage issues its E loads
// together (term loop outermost) -- a per-coefficient evaluation would serialise one memory latency per term.
template <int E>
__device__ __forceinline__ void rs_operand(u64 (&v)[E], const int (&g)[E], const RsItem &it, const SumSrc *__restrict__ srcs, int p, int i,
                                           size_t N, const DModulus &M)
{
    if (it.count == 0) {
        const u64 *x = it.src.limb(p, i, N);
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = x[g[j]];
    } else {
        u64 s[E];
        Acc128 a[E];
#pragma unroll
        for (int j = 0; j < E; j++) s[j] = 0, a[j].clear();
        int n_plain = 0, n_prod = 0;
        for (int t = 0; t < it.count; t++) {
            const SumSrc src = srcs[it.first + t];
            const u64 *x = src.v.limb(p, i, N);
            if (src.plain) {
                const u64 *w = src.plain + (size_t)i * N;
#pragma unroll
                for (int j = 0; j < E; j++) a[j].mac(x[g[j]], w[g[j]]);
                if ((++n_prod & 15) == 0) {
#pragma unroll
                    for (int j = 0; j < E; j++) s[j] = fold60(s[j], M.delta) + a[j].reduce(M), a[j].clear();
                }
            } else {
#pragma unroll
                for (int j = 0; j < E; j++) s[j] += x[g[j]];
                if ((++n_plain & 7) == 0) {
#pragma unroll
                    for (int j = 0; j < E; j++) s[j] = fold60(s[j], M.delta);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = addmod(canon(s[j], M), a[j].reduce(M), M.q);
    }
    if (it.add && p == 0) {
        const u64 *w = it.add + (size_t)i * N;
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = addmod(v[j], w[g[j]], M.q);
    }
    if (it.mul) {
        const u64 *w = it.mul + (size_t)i * N;
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = mulmod(v[j], w[g[j]], M);
    }
}

// opcode 10, first launch: Decryptor::decrypt (c0 + c1*s) of every item's operand fused into the first inverse phase.
// z = b*ell + i.  The operand may be an expression (a rescale folded into the opcode, with what had been folded into it).
template <int K, int LOGE>
__global__ __launch_bounds__(kTileThreads) void f_irows_boot_kernel(const BootItem *__restrict__ items, const SumSrc *__restrict__ srcs,
                                                                     const u64 *__restrict__ sk, int ell, u64 *__restrict__ out,
                                                                     const DModulus *__restrict__ mods, const u64 *__restrict__ itw,
                                                                     int logN)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    constexpr int E = 1 << LOGE;
    const int z = blockIdx.y, i = z % ell;
    const size_t N = (size_t)1 << logN;
    const BootItem bi = items[z / ell];
    const RsItem it{ bi.src, bi.dst, bi.first, bi.count, bi.add, bi.mul };
    const DModulus M = mods[i];
    const u64 *s = sk + (size_t)i * N;
    u64 *o = out + (size_t)z * N;
    int g[E];
#pragma unroll
    for (int j = 0; j < E; j++) g[j] = tile_gidx<K, LOGE, false>(num_passes<LOGE>(K) - 1, logN, blockIdx.x, j);
    u64 x[E], x1[E];
    rs_operand<E>(x, g, it, srcs, 0, i, N, M);
    rs_operand<E>(x1, g, it, srcs, 1, i, N, M);
#pragma unroll
    for (int j = 0; j < E; j++) x[j] = addmod(x[j], mulmod(x1[j], s[g[j]], M), M.q);
    auto nold = [](int) -> u64 { return 0; };
    ntt_tile_x<K, LOGE, false, true, false, true, false>(x, M, itw + ((size_t)i << logN), logN, blockIdx.x, nold,
                                                         [=](int gi, u64 v) { o[gi] = v; }, lds);
}

// R1: inverse ROWS phase of the limb a rescale drops.  z = b*2 + p
template <int K, int LOGE>
__global__ __launch_bounds__(kTileThreads) void f_irows_rs_kernel(const RsItem *__restrict__ items, const SumSrc *__restrict__ srcs, int l,
                                                                   u64 *__restrict__ out, const DModulus *__restrict__ mods,
                                                                   const u64 *__restrict__ itw, int logN)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    constexpr int E = 1 << LOGE;
    const int z = blockIdx.y;
    const size_t N = (size_t)1 << logN;
    const DModulus M = mods[l];
    u64 *o = out + (size_t)z * N;
    int g[E];
#pragma unroll
    for (int j = 0; j < E; j++) g[j] = tile_gidx<K, LOGE, false>(num_passes<LOGE>(K) - 1, logN, blockIdx.x, j);
    u64 x[E];
    rs_operand<E>(x, g, items[z >> 1], srcs, z & 1, l, N, M);
    auto nold = [](int) -> u64 { return 0; };
    ntt_tile_x<K, LOGE, false, true, false, true, false>(x, M, itw + ((size_t)l << logN), logN, blockIdx.x, nold,
                                                         [=](int gi, u64 v) { o[gi] = v; }, lds);
}

struct SrcRsLast1 { // single rescale without a device item table (encryption's divide-and-round)
    CtView src;
    int l;
    __device__ int prime(int) const { return l; }
    __device__ u64 load(int z, int g, int logN) const { return src.limb(z & 1, l, (size_t)1 << logN)[g]; }
};
struct SrcDecrypt { // limb z of c0 + c1*s  (Decryptor::decrypt fused into the first inverse phase)
    CtView ct;
    const u64 *sk;
    const DModulus *mods;
    __device__ int prime(int z) const { return z; }
    __device__ u64 load(int z, int g, int logN) const
    {
        const size_t N = (size_t)1 << logN;
        const DModulus M = mods[z];
        return addmod(ct.limb(0, z, N)[g], mulmod(ct.limb(1, z, N)[g], sk[(size_t)z * N + g], M), M.q);
    }
};

struct SrcTensorC2 { // limb z = b*ell + i of a1*b1, the c2 of item b's tensor product (ckks_multiply) -- never materialised
    const MulItem *items;
    const DModulus *mods;
    int ell;
    __device__ int prime(int z) const { return z % ell; }
    __device__ u64 load(int z, int g, int logN) const
    {
        const size_t N = (size_t)1 << logN;
        const int i = z % ell;
        const MulItem &it = items[z / ell];
        return mulmod(it.a.limb(1, i, N)[g], it.b.limb(1, i, N)[g], mods[i]);
    }
};
// L1 / L5 / R1: inverse ROWS phase, out[z] (lazy values) = phase(src limb z)
template <int K, int LOGE, class Src>
__global__ __launch_bounds__(kTileThreads) void f_irows_kernel(Src src, u64 *__restrict__ out, long out_stride,
                                                                const DModulus *__restrict__ mods, const u64 *__restrict__ itw,
                                                                int logN)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    const int z = blockIdx.y, p = src.prime(z);
    u64 *o = out + (long)z * out_stride;
    ntt_tile<K, LOGE, false, true, false>(
        mods[p], itw + ((size_t)p << logN), logN, blockIdx.x, [=](int g) { return src.load(z, g, logN); },
        [=](int g, u64 v) { o[g] = v; }, lds);
}

// L2: z = (b*l + j)*l + e
// MERGE (throughput-bound launches, grid.y = B*l): one workgroup finishes the iNTT of digit (b, j) ONCE and runs the base change + first
// NTT phase for each of its l target moduli in turn, instead of l workgroups each recomputing the inverse phase (a quarter of the
// launch's work at l = 2, a third at l = 3) -- what the large-batch path gets from two launches and a round trip through HBM.
template <int K, int LOGE, bool MERGE>
__global__ __launch_bounds__(kTileThreads) void f_ks_icols_lift_fcols_kernel(const u64 *__restrict__ digits, u64 *__restrict__ ext,
                                                                              int ell, int sp, const DModulus *__restrict__ mods,
                                                                              const u64 *__restrict__ tw, const u64 *__restrict__ itw,
                                                                              int logN, const u64 *__restrict__ tw2)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    const int z = blockIdx.y, dj = MERGE ? z : z / ell, j = dj % ell;
    const size_t N = (size_t)1 << logN;
    const u64 *in = digits + (size_t)dj * N;
    u64 x[1 << LOGE];
    auto nost = [](int, u64) {};
    ntt_tile_x<K, LOGE, true, true, true, false, true>(
        x, mods[j], itw + ((size_t)j << logN), logN, blockIdx.x, [=](int g) { return in[g]; }, nost, lds);
    auto nold = [](int) -> u64 { return 0; };
    for (int e = MERGE ? 0 : z % ell; e < (MERGE ? ell : z % ell + 1); e++) {
        u64 *out = ext + ((size_t)dj * ell + e) * N;
        const int pm = ks_other_prime(j, e, ell, sp);
        const DModulus Mm = mods[pm];
        u64 y[1 << LOGE];
#pragma unroll
        for (int r = 0; r < (1 << LOGE); r++) y[r] = recanon(x[r], Mm); // one conditional subtraction within a width class (modarith.hpp)
        __syncthreads(); // the previous tile's last LDS image has been read by everyone
        ntt_tile_fcols<K, LOGE, false, true, false>(
            y, Mm, tw + ((size_t)pm << logN), tw2 ? tw2 + ((size_t)pm << (K + 1)) : nullptr, logN, blockIdx.x, nold, [=](int g, u64 v) { out[g] = v; }, lds);
    }
}

// L6 / R2: z = bp*cnt + i : finish the iNTT of the dropped limb bp (prime l), round, change base to prime i, first NTT phase
// MERGE (grid.y = polys): the same sharing as in L2 -- the dropped limb's inverse phase once, then every target modulus in turn
template <int K, int LOGE, bool MERGE>
__global__ __launch_bounds__(kTileThreads) void f_dr_icols_lift_fcols_kernel(const u64 *__restrict__ last, long last_stride,
                                                                              u64 *__restrict__ tmp, int cnt, int l, int Kp,
                                                                              const DModulus *__restrict__ mods,
                                                                              const u64 *__restrict__ half_mod,
                                                                              const u64 *__restrict__ tw, const u64 *__restrict__ itw,
                                                                              int logN, const u64 *__restrict__ tw2)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    const int z = blockIdx.y, bp = MERGE ? z : z / cnt;
    const size_t N = (size_t)1 << logN;
    const u64 *in = last + (long)bp * last_stride;
    u64 x[1 << LOGE];
    auto nost = [](int, u64) {};
    ntt_tile_x<K, LOGE, true, true, true, false, true>(
        x, mods[l], itw + ((size_t)l << logN), logN, blockIdx.x, [=](int g) { return in[g]; }, nost, lds);
    const u64 ql = mods[l].q, half = ql >> 1;
#pragma unroll
    for (int r = 0; r < (1 << LOGE); r++) { // RNSTool::divide_and_round_q_last_ntt_inplace, coefficient-domain part: + floor(q_l / 2) mod q_l
        const u64 y = x[r] + half;
        x[r] = y >= ql ? y - ql : y;
    }
    auto nold = [](int) -> u64 { return 0; };
    for (int i = MERGE ? 0 : z % cnt; i < (MERGE ? cnt : z % cnt + 1); i++) {
        u64 *out = tmp + ((size_t)bp * cnt + i) * N;
        const DModulus Mi = mods[i];
        const u64 qi = Mi.q, neg_half = qi - half_mod[(size_t)l * Kp + i];
        u64 y[1 << LOGE];
#pragma unroll
        for (int r = 0; r < (1 << LOGE); r++) { // ... reduced into q_i, - floor(q_l / 2) mod q_i
            const u64 v = recanon(x[r], Mi) + neg_half;
            y[r] = v >= qi ? v - qi : v;
        }
        __syncthreads();
        ntt_tile_fcols<K, LOGE, false, true, false>(
            y, Mi, tw + ((size_t)i << logN), tw2 ? tw2 + ((size_t)i << (K + 1)) : nullptr, logN, blockIdx.x, nold, [=](int g, u64 v) { out[g] = v; }, lds);
    }
}

// Throughput variants of L2 / L6 for large batches: the inverse COLS phase has already been run once per source limb
// (launch_ntt_cols_inv, canonical output), so each workgroup only does the base change and ONE forward phase instead of
// recomputing the inverse phase per target modulus.
template <int K, int LOGE>
__global__ __launch_bounds__(kTileThreads) void f_ks_lift_fcols_kernel(const u64 *__restrict__ digits, u64 *__restrict__ ext, int ell,
                                                                        int sp, const DModulus *__restrict__ mods,
                                                                        const u64 *__restrict__ tw, int logN, const u64 *__restrict__ tw2)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    const int z = blockIdx.y, e = z % ell, dj = z / ell, j = dj % ell;
    const size_t N = (size_t)1 << logN;
    const u64 *in = digits + (size_t)dj * N;
    u64 *out = ext + (size_t)z * N;
    const int pm = ks_other_prime(j, e, ell, sp);
    const DModulus Mm = mods[pm];
    u64 x[1 << LOGE];
    ntt_tile_fcols<K, LOGE, false, false, false>(
        x, Mm, tw + ((size_t)pm << logN), tw2 ? tw2 + ((size_t)pm << (K + 1)) : nullptr, logN, blockIdx.x,
        [=](int g) {
            return recanon(in[g], Mm);
        },
        [=](int g, u64 v) { out[g] = v; }, lds);
}

template <int K, int LOGE>
__global__ __launch_bounds__(kTileThreads) void f_dr_lift_fcols_kernel(const u64 *__restrict__ last, long last_stride,
                                                                        u64 *__restrict__ tmp, int cnt, int l, int Kp,
                                                                        const DModulus *__restrict__ mods,
                                                                        const u64 *__restrict__ half_mod, const u64 *__restrict__ tw,
                                                                        int logN, const u64 *__restrict__ tw2)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    const int z = blockIdx.y, i = z % cnt, bp = z / cnt;
    const size_t N = (size_t)1 << logN;
    const u64 *in = last + (long)bp * last_stride;
    u64 *out = tmp + (size_t)z * N;
    const DModulus Mi = mods[i];
    const u64 ql = mods[l].q, qi = Mi.q, half = ql >> 1;
    const u64 neg_half = qi - half_mod[(size_t)l * Kp + i];
    u64 x[1 << LOGE];
    ntt_tile_fcols<K, LOGE, false, false, false>(
        x, Mi, tw + ((size_t)i << logN), tw2 ? tw2 + ((size_t)i << (K + 1)) : nullptr, logN, blockIdx.x,
        [=](int g) {
            u64 y = in[g] + half;
            y = y >= ql ? y - ql : y;
            y = recanon(y, Mi);
            y += neg_half;
            return y >= qi ? y - qi : y;
        },
        [=](int g, u64 v) { out[g] = v; }, lds);
}

// L7 / R3: z = bp*cnt + i.  MODE 0 rotation, 1 relinearisation, 2 rescale
template <int K, int LOGE, int MODE>
__global__ __launch_bounds__(kTileThreads) void f_frows_final_kernel(const u64 *__restrict__ tmp, const void *__restrict__ items_,
                                                                      RsItem single, const u64 *__restrict__ plain,
                                                                      const SumSrc *__restrict__ srcs,
                                                                      const u64 *__restrict__ acc, int cnt, int l, int Kp,
                                                                      const DModulus *__restrict__ mods,
                                                                      const u64 *__restrict__ inv_last, const u64 *__restrict__ tw,
                                                                      int logN, int base_folded)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    const int z = blockIdx.y, i = z % cnt, bp = z / cnt, b = bp >> 1, p = bp & 1;
    const size_t N = (size_t)1 << logN;
    const DModulus M = mods[i];
    const u64 inv = inv_last[(size_t)l * Kp + i];
    const u64 *in = tmp + (size_t)z * N;
    if (MODE == 0 && base_folded) {
        // The rotation's base term galois(c0) already rides on the accumulator (f_ks_frows_mac_kernel added P galois(c0) to it:
        // galois(c0) + (acc - t) P^-1 = ((acc + P galois(c0)) - t) P^-1 exactly), so this kernel has no gather: a ROWS tile leaves a thread
        // 2^LOGE CONSECUTIVE coefficients and the epilogue runs on 16-byte vectors.  (The per-coefficient store functor below -- 8-byte accesses
        // at a 2^LOGE-word lane stride, a Galois gather with 64 lanes in 64 cache lines -- is what hybrid_fused.hip's last kernel measured 2.5x
        // slower than this form.)
        constexpr int E = 1 << LOGE;
        const KsItem it = reinterpret_cast<const KsItem *>(items_)[b];
        const u64 *x = acc + (((size_t)bp) * (cnt + 1) + i) * N;
        u64 *o = it.dst.limb(p, i, N);
        u64 v[E];
        auto nost = [](int, u64) {};
        ntt_tile_x<K, LOGE, false, false, true, false, true>(v, M, tw + ((size_t)i << logN), logN, blockIdx.x, [=](int gi) { return in[gi]; }, nost, lds);
        const int g0 = tile_gidx<K, LOGE, false>(num_passes<LOGE>(K) - 1, logN, blockIdx.x, 0);
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < E; j += 2) {
            const u64x2 a = *reinterpret_cast<const u64x2 *>(x + g0 + j);
            u64x2 r;
            r.x = mulmod(submod(a.x, v[j], M.q), inv, M), r.y = mulmod(submod(a.y, v[j + 1], M.q), inv, M);
            *reinterpret_cast<u64x2 *>(o + g0 + j) = r;
        }
    } else if (MODE == 0) {
        const KsItem it = reinterpret_cast<const KsItem *>(items_)[b];
        const u64 *x = acc + (((size_t)bp) * (cnt + 1) + i) * N;
        const u64 *c0 = it.src.limb(0, i, N);
        u64 *o = it.dst.limb(p, i, N);
        ntt_tile<K, LOGE, false, false, true>(
            M, tw + ((size_t)i << logN), logN, blockIdx.x, [=](int g) { return in[g]; },
            [=](int g, u64 v) {
                const u64 base = p == 0 ? c0[galois_idx((u32)g, it.elt, logN)] : 0;
                o[g] = addmod(base, mulmod(submod(x[g], v, M.q), inv, M), M.q);
            },
            lds);
    } else if (MODE == 1) {
        const MulItem it = reinterpret_cast<const MulItem *>(items_)[b];
        const u64 *x = acc + (((size_t)bp) * (cnt + 1) + i) * N;
        u64 *o = it.dst.limb(p, i, N);
        ntt_tile<K, LOGE, false, false, true>(
            M, tw + ((size_t)i << logN), logN, blockIdx.x, [=](int g) { return in[g]; },
            [=](int g, u64 v) { o[g] = addmod(o[g], mulmod(submod(x[g], v, M.q), inv, M), M.q); }, lds);
    } else if (MODE == 4) { // relinearisation whose c0, c1 tensor terms are computed here (dst holds nothing yet)
        const MulItem it = reinterpret_cast<const MulItem *>(items_)[b];
        const u64 *x = acc + (((size_t)bp) * (cnt + 1) + i) * N;
        const u64 *a0 = it.a.limb(0, i, N), *a1 = it.a.limb(1, i, N), *b0 = it.b.limb(0, i, N), *b1 = it.b.limb(1, i, N);
        u64 *o = it.dst.limb(p, i, N);
        constexpr int E = 1 << LOGE;
        u64 v[E], base[E];
        int g[E];
        auto nost = [](int, u64) {};
        ntt_tile_x<K, LOGE, false, false, true, false, true>(v, M, tw + ((size_t)i << logN), logN, blockIdx.x,
                                                             [=](int gi) { return in[gi]; }, nost, lds);
#pragma unroll
        for (int j = 0; j < E; j++) g[j] = tile_gidx<K, LOGE, false>(num_passes<LOGE>(K) - 1, logN, blockIdx.x, j);
        if (p == 0) {
#pragma unroll
            for (int j = 0; j < E; j++) base[j] = mulmod(a0[g[j]], b0[g[j]], M);
        } else {
#pragma unroll
            for (int j = 0; j < E; j++) {
                Acc128 t;
                t.clear();
                t.mac(a0[g[j]], b1[g[j]]);
                t.mac(a1[g[j]], b0[g[j]]);
                base[j] = t.reduce(M);
            }
        }
#pragma unroll
        for (int j = 0; j < E; j++) o[g[j]] = addmod(base[j], mulmod(submod(x[g[j]], v[j], M.q), inv, M), M.q);
    } else if (MODE == 3) { // one rescale given by value, optionally followed by "+ plaintext" on c0 (Encryptor::encrypt)
        const u64 *x = single.src.limb(p, i, N);
        u64 *o = single.dst.limb(p, i, N);
        const u64 *pl = (plain && p == 0) ? plain + (size_t)i * N : nullptr;
        ntt_tile<K, LOGE, false, false, true>(
            M, tw + ((size_t)i << logN), logN, blockIdx.x, [=](int g) { return in[g]; },
            [=](int g, u64 v) {
                const u64 r = mulmod(submod(x[g], v, M.q), inv, M);
                o[g] = pl ? addmod(r, pl[g], M.q) : r;
            },
            lds);
    } else {
        const RsItem it = reinterpret_cast<const RsItem *>(items_)[b];
        u64 *o = it.dst.limb(p, i, N);
        constexpr int E = 1 << LOGE;
        u64 x[E], op[E];
        int g[E];
        auto nost = [](int, u64) {};
        ntt_tile_x<K, LOGE, false, false, true, false, true>(x, M, tw + ((size_t)i << logN), logN, blockIdx.x,
                                                             [=](int gi) { return in[gi]; }, nost, lds);
#pragma unroll
        for (int j = 0; j < E; j++) g[j] = tile_gidx<K, LOGE, false>(num_passes<LOGE>(K) - 1, logN, blockIdx.x, j);
        rs_operand<E>(op, g, it, srcs, p, i, N, M);
#pragma unroll
        for (int j = 0; j < E; j++) o[g[j]] = mulmod(submod(op[j], x[j], M.q), inv, M);
    }
}

// ---- L7 / R3 with a continuation into the consumer's first phase (plan.hpp Handoff) ---------------------------------------------------
// Output polynomial p, limb i of item b at this tile's coefficients g[] (pass NP-1 layout of a ROWS tile): computed exactly as
// f_frows_final_kernel does for the same MODE, stored to the item's destination, and left in v[] (canonical).
template <int K, int LOGE, int MODE>
__device__ __forceinline__ void final_value(u64 (&v)[1 << LOGE], const int (&g)[1 << LOGE], const u64 *__restrict__ in, const void *__restrict__ items_,
                                            const SumSrc *__restrict__ srcs, const u64 *__restrict__ acc, int b, int p, int i, int cnt,
                                            const DModulus &M, u64 inv, const u64 *__restrict__ tw, int logN, u64 *__restrict__ lds,
                                            bool base_folded = false)
{
    constexpr int E = 1 << LOGE;
    const size_t N = (size_t)1 << logN;
    auto nost = [](int, u64) {};
    u64 x[E];
    ntt_tile_x<K, LOGE, false, false, true, false, true>(x, M, tw + ((size_t)i << logN), logN, blockIdx.x, [=](int gi) { return in[gi]; }, nost, lds);
    if (MODE == 4) {
        const MulItem it = reinterpret_cast<const MulItem *>(items_)[b];
        const u64 *ac = acc + (((size_t)(b * 2 + p)) * (cnt + 1) + i) * N;
        const u64 *a0 = it.a.limb(0, i, N), *a1 = it.a.limb(1, i, N), *b0 = it.b.limb(0, i, N), *b1 = it.b.limb(1, i, N);
        u64 *o = it.dst.limb(p, i, N);
#pragma unroll
        for (int j = 0; j < E; j++) {
            u64 base;
            if (p == 0)
                base = mulmod(a0[g[j]], b0[g[j]], M);
            else {
                Acc128 t;
                t.clear();
                t.mac(a0[g[j]], b1[g[j]]);
                t.mac(a1[g[j]], b0[g[j]]);
                base = t.reduce(M);
            }
            v[j] = addmod(base, mulmod(submod(ac[g[j]], x[j], M.q), inv, M), M.q);
            o[g[j]] = v[j];
        }
    } else if (MODE == 0) {
        const KsItem it = reinterpret_cast<const KsItem *>(items_)[b];
        const u64 *ac = acc + (((size_t)(b * 2 + p)) * (cnt + 1) + i) * N;
        const u64 *c0 = it.src.limb(0, i, N);
        u64 *o = it.dst.limb(p, i, N);
#pragma unroll
        for (int j = 0; j < E; j++) {
            const u64 base = (p == 0 && !base_folded) ? c0[galois_idx((u32)g[j], it.elt, logN)] : 0; // (folded: the accumulator carries it)
            v[j] = addmod(base, mulmod(submod(ac[g[j]], x[j], M.q), inv, M), M.q);
            o[g[j]] = v[j];
        }
    } else { // MODE 2: rescale
        const RsItem it = reinterpret_cast<const RsItem *>(items_)[b];
        u64 *o = it.dst.limb(p, i, N);
        u64 op[E];
        rs_operand<E>(op, g, it, srcs, p, i, N, M);
#pragma unroll
        for (int j = 0; j < E; j++) {
            v[j] = mulmod(submod(op[j], x[j], M.q), inv, M);
            o[g[j]] = v[j];
        }
    }
}

// CONT_RS / CONT_MUL: grid = (tiles, 2B * cnt), z = (b*2 + p)*cnt + i as in f_frows_final_kernel; the workgroups that hold what the
// consumer's first phase reads carry on.  CONT_BOOT: grid = (tiles, B * cnt), z = b*cnt + i, both polynomials in one workgroup
// (the consumer decrypts: c0 + c1*s).
template <int K, int LOGE, int MODE, int CONT>
__global__ __launch_bounds__(kTileThreads) void f_frows_final_cont_kernel(const u64 *__restrict__ tmp, const void *__restrict__ items_,
                                                                           const SumSrc *__restrict__ srcs, const u64 *__restrict__ acc, int cnt,
                                                                           int l, int Kp, const DModulus *__restrict__ mods,
                                                                           const u64 *__restrict__ inv_last, const u64 *__restrict__ tw,
                                                                           const u64 *__restrict__ itw, int logN, Handoff h, int base_folded)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    constexpr int E = 1 << LOGE, NP = num_passes<LOGE>(K);
    const size_t N = (size_t)1 << logN;
    int g[E];
#pragma unroll
    for (int j = 0; j < E; j++) g[j] = tile_gidx<K, LOGE, false>(NP - 1, logN, blockIdx.x, j);
    auto nold = [](int) -> u64 { return 0; };
    if (CONT == CONT_BOOT) {
        const int z = blockIdx.y, i = z % cnt, b = z / cnt;
        const DModulus M = mods[i];
        const u64 inv = inv_last[(size_t)l * Kp + i];
        u64 v0[E], v1[E];
        final_value<K, LOGE, MODE>(v0, g, tmp + ((size_t)(b * 2 + 0) * cnt + i) * N, items_, srcs, acc, b, 0, i, cnt, M, inv, tw, logN, lds, base_folded != 0);
        __syncthreads(); // the tile's last LDS image has been read by everyone
        final_value<K, LOGE, MODE>(v1, g, tmp + ((size_t)(b * 2 + 1) * cnt + i) * N, items_, srcs, acc, b, 1, i, cnt, M, inv, tw, logN, lds, base_folded != 0);
        const u64 *sk = h.sk + (size_t)i * N;
#pragma unroll
        for (int j = 0; j < E; j++) v0[j] = addmod(v0[j], mulmod(v1[j], sk[g[j]], M), M.q);
        __syncthreads();
        u64 *o = h.out + (size_t)z * N;
        ntt_tile_x<K, LOGE, false, true, false, true, false>(v0, M, itw + ((size_t)i << logN), logN, blockIdx.x, nold, [=](int gi, u64 v) { o[gi] = v; }, lds);
    } else {
        const int z = blockIdx.y, i = z % cnt, bp = z / cnt, b = bp >> 1, p = bp & 1;
        const DModulus M = mods[i];
        const u64 inv = inv_last[(size_t)l * Kp + i];
        u64 v[E];
        final_value<K, LOGE, MODE>(v, g, tmp + (size_t)z * N, items_, srcs, acc, b, p, i, cnt, M, inv, tw, logN, lds, base_folded != 0);
        if (CONT == CONT_RS) {
            if (i != cnt - 1) return; // the consumer drops the last limb: only its workgroups continue (uniform per workgroup)
            const RsItem cit = h.rs_items[b]; // single-use "+ plaintext" / "* plaintext" folded into the consumer (rs_operand's order)
            if (cit.add && p == 0) {
                const u64 *w = cit.add + (size_t)i * N;
#pragma unroll
                for (int j = 0; j < E; j++) v[j] = addmod(v[j], w[g[j]], M.q);
            }
            if (cit.mul) {
                const u64 *w = cit.mul + (size_t)i * N;
#pragma unroll
                for (int j = 0; j < E; j++) v[j] = mulmod(v[j], w[g[j]], M);
            }
            __syncthreads();
            u64 *o = h.out + (size_t)bp * N;
            ntt_tile_x<K, LOGE, false, true, false, true, false>(v, M, itw + ((size_t)i << logN), logN, blockIdx.x, nold, [=](int gi, u64 y) { o[gi] = y; }, lds);
        } else { // CONT_MUL
            if (p != 1) return;
            const CtView other = h.other[b];
            if (other.p) {
                const u64 *o1 = other.limb(1, i, N);
#pragma unroll
                for (int j = 0; j < E; j++) v[j] = mulmod(v[j], o1[g[j]], M);
            } else {
#pragma unroll
                for (int j = 0; j < E; j++) v[j] = mulmod(v[j], v[j], M);
            }
            __syncthreads();
            u64 *o = h.out + ((size_t)b * cnt + i) * N;
            ntt_tile_x<K, LOGE, false, true, false, true, false>(v, M, itw + ((size_t)i << logN), logN, blockIdx.x, nold, [=](int gi, u64 y) { o[gi] = y; }, lds);
        }
    }
}

// L3 + L4 + L5 in one launch (latency path): grid = (tiles, l + 2, B).  Workgroup (tile, y, b) owns one ROWS-phase tile of
// output modulus slot m (y < l: prime y, both accumulators; y = l, l + 1: the special prime, accumulator y - l).  For every
// digit j it finishes the NTT of the lifted digit (forward ROWS phase, result kept in registers) -- or, for j == m, reads the
// operand itself, which is already in NTT form -- and multiplies by the key limb at the same coefficients.  The special-prime
// accumulators continue in registers into the inverse ROWS phase that the mod-down starts with.
// MODE 0: rotation (operand = c1 of the item through its Galois permutation, key per item); MODE 1: relinearisation.
template <int K, int LOGE, int MODE, bool MERGE>
__global__ __launch_bounds__(kTileThreads) void f_ks_frows_mac_kernel(const u64 *__restrict__ ext, const u64 *__restrict__ target,
                                                                       const KsItem *__restrict__ items,
                                                                       const u64 *__restrict__ shared_key, u64 *__restrict__ acc, int ell,
                                                                       int Kp, const DModulus *__restrict__ mods,
                                                                       const u64 *__restrict__ tw, const u64 *__restrict__ itw, int logN,
                                                                       const u64 *__restrict__ pmod, int items_fast)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    constexpr int E = 1 << LOGE, NP = num_passes<LOGE>(K);
    // items_fast: grid = (tiles, B, rows) instead of (tiles, rows, B).  The workgroups (tile, row) of consecutive items are then 2^k tiles apart
    // in launch order -- the same XCD (workgroup w runs on XCD w mod 8), dispatched together -- and items that use the same key (the plan
    // sorts a rotation step's items by Galois element; a relinearisation step has one key) read each key tile out of that XCD's L2 after
    // the first of them fetched it.  SEAL's default key set has 28 elements, so a 64-item step of a convolution names each key several times:
    // with the rows slower than the items, two readers of a key tile were a whole item (~150 MB of traffic at 13 primes) apart.
    const int y = items_fast ? blockIdx.z : blockIdx.y, b = items_fast ? blockIdx.y : blockIdx.z, sp = Kp - 1;
    // psel < 0: both accumulators.  MERGE (grid.y = l + 1, throughput-bound launches): ONE workgroup row does the special prime for both
    // accumulators -- the l transforms of the lifted digits once instead of twice, then the two inverse ROWS phases one after the other
    // (its own instantiation: both accumulators live through the epilogue cost 16-20 VGPRs, a wave per SIMD)
    const int m = y < ell ? y : ell, psel = (MERGE && y == ell) ? -1 : y - ell;
    const int pm = m == ell ? sp : m;
    const size_t N = (size_t)1 << logN;
    const DModulus M = mods[pm];
    const u64 *key = MODE == 0 ? items[b].key : shared_key;
    int g[E];
#pragma unroll
    for (int e = 0; e < E; e++) g[e] = tile_gidx<K, LOGE, false>(NP - 1, logN, blockIdx.x, e);
    Acc128 a0[E], a1[E];
#pragma unroll
    for (int e = 0; e < E; e++) a0[e].clear(), a1[e].clear();
    if (MODE == 0 && pmod && m < ell) {
        // A rotation's base term rides on the accumulator: galois(c0) + (acc - t) P^-1 = ((acc + P galois(c0)) - t) P^-1 exactly, so the last
        // kernel of the key switch has no gather to do (this one has the Galois map in hand for c1 anyway).  The residue P galois(c0) mod q
        // is the accumulator's START value -- below 2^60 like a folded window, so the 16-products-per-window bound is unchanged -- and it is
        // computed here, before anything else is live (added in the epilogue it cost the merged form 6 VGPRs and a wave per SIMD).
        const KsItem &it = items[b];
        const u64 *c0 = it.src.limb(0, m, N);
        const u64 P = pmod[m];
        u64 cv[E];
        galois_gather<E>(cv, c0, (u32)g[0], it.elt, logN);
#pragma unroll
        for (int e = 0; e < E; e++) a0[e].lo = mulmod(cv[e], P, M);
    }
    auto nost = [](int, u64) {};
    auto nold = [](int) -> u64 { return 0; };
    bool lds_used = false;
    // Latency-shaped launches (not MERGE; one or two workgroups per CU, each walking its digits serially): the walk is software-pipelined.
    //   * the tile's twiddles are the same for every digit (same prime, same tile): fetched once (tile_twiddles), not once per transform
    //   * digit j + 1's coefficients and digit j's key limbs are requested BEFORE digit j's transform: the loads fly under its passes
    // so that a digit costs its butterflies and LDS exchanges, not a memory round trip before and another after them.  The extra live
    // registers (twiddles, next coefficients, key limbs) do not cost occupancy where this form runs: such launches have at most two waves per SIMD.
    constexpr bool PF = !MERGE && LOGE <= 2;
    u64 wtw[PF ? NP : 1][E];
    // a ROWS tile's first forward pass holds coefficients gin0 + e * SUBT, its last pass leaves g[0] + e (PassMap::idx_of)
    constexpr int SUBT = (1 << K) >> LOGE;
    int gin0 = 0;
    if (PF) {
        tile_twiddles<K, PF ? LOGE : 1, false, false>(reinterpret_cast<u64 (&)[num_passes<PF ? LOGE : 1>(K)][1 << (PF ? LOGE : 1)]>(wtw), tw + ((size_t)pm << logN),
                                                      logN, blockIdx.x);
        gin0 = tile_gidx<K, LOGE, false>(0, logN, blockIdx.x, 0);
    }
    // digit j's operand at this tile: its own limb (j == m: NTT form already, thread <-> coefficient map g) or the raised limb's first-pass image
    auto own_on_the_fly = [&](int j) { return j == m && MODE != 0 && !target; };
    auto fetch_x = [&](int j, u64 (&x)[E]) {
        const bool own = j == m;
        if (MODE == 0 && own) {
            const KsItem &it = items[b];
            galois_gather<E>(x, it.src.limb(1, j, N), (u32)g[0], it.elt, logN);
        } else if (!own || target) { // (one load path with a selected base and stride: two index arrays made the compiler select between them in memory)
            const u64 *src = own ? target + ((size_t)b * ell + j) * N : ext + (((size_t)b * ell + j) * ell + (m < j ? m : m - 1)) * N;
            const int base = own ? g[0] : gin0, stride = own ? 1 : SUBT;
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = src[base + e * stride];
        }
    };
    // (round 5, measured and not kept: TWO digits ahead -- digit j + 2's coefficients and digit j + 1's key limbs requested before digit j's
    // transform, +24 VGPRs -- does nothing for a 13-prime hop (84.9 against 84.6-86.1 us) and costs config 3 a workgroup per CU: its 1 664
    // workgroups at 178 VGPRs run 2 to a CU instead of 3, 459 against 430-443 us; profiles/r05_experiments.txt)
    u64 xn[E];
    if (PF) fetch_x(0, xn);
    for (int j = 0; j < ell; j++) {
        u64 x[E];
        const u64 *k0 = key + (((size_t)j * 2 + 0) * Kp + pm) * N, *k1 = key + (((size_t)j * 2 + 1) * Kp + pm) * N;
        u64 kv0[E], kv1[E];
        if (PF && psel != 1) {
#pragma unroll
            for (int e = 0; e < E; e++) kv0[e] = k0[g[e]];
        }
        if (PF && psel != 0) {
#pragma unroll
            for (int e = 0; e < E; e++) kv1[e] = k1[g[e]];
        }
        if (PF) {
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = xn[e];
            if (j + 1 < ell) fetch_x(j + 1, xn);
            if (own_on_the_fly(j)) { // c2 = a1*b1 on the fly (items points at the MulItem table)
                const MulItem &it = reinterpret_cast<const MulItem *>(items)[b];
                const u64 *a1 = it.a.limb(1, j, N), *b1 = it.b.limb(1, j, N);
#pragma unroll
                for (int e = 0; e < E; e++) x[e] = mulmod(a1[g[e]], b1[g[e]], M);
            } else if (j != m) {
                if (lds_used) __syncthreads(); // the previous tile's last LDS image has been read by everyone
                ntt_tile_x<K, LOGE, false, false, true, true, true>(x, M, tw + ((size_t)pm << logN), logN, blockIdx.x, nold, nost, lds,
                                                                    PF ? wtw : nullptr);
                lds_used = true;
            }
        } else if (j == m) {
            if (MODE == 0) {
                const KsItem &it = items[b];
                galois_gather<E>(x, it.src.limb(1, j, N), (u32)g[0], it.elt, logN);
            } else if (target) {
                const u64 *tg = target + ((size_t)b * ell + j) * N;
#pragma unroll
                for (int e = 0; e < E; e++) x[e] = tg[g[e]];
            } else { // c2 = a1*b1 on the fly (items points at the MulItem table)
                const MulItem &it = reinterpret_cast<const MulItem *>(items)[b];
                const u64 *a1 = it.a.limb(1, j, N), *b1 = it.b.limb(1, j, N);
#pragma unroll
                for (int e = 0; e < E; e++) x[e] = mulmod(a1[g[e]], b1[g[e]], M);
            }
        } else {
            const u64 *in = ext + (((size_t)b * ell + j) * ell + (m < j ? m : m - 1)) * N;
            if (lds_used) __syncthreads(); // the previous tile's last LDS image has been read by everyone
            ntt_tile_x<K, LOGE, false, false, true, false, true>(x, M, tw + ((size_t)pm << logN), logN, blockIdx.x,
                                                                 [=](int gi) { return in[gi]; }, nost, lds);
            lds_used = true;
        }
        if (psel != 1) {
#pragma unroll
            for (int e = 0; e < E; e++) a0[e].mac(x[e], PF ? kv0[e] : k0[g[e]]);
        }
        if (psel != 0) {
#pragma unroll
            for (int e = 0; e < E; e++) a1[e].mac(x[e], PF ? kv1[e] : k1[g[e]]);
        }
        if ((j & 15) == 15 && j + 1 < ell) { // a 128-bit accumulator holds 16 products of canonical residues (Acc128): fold it into a word
#pragma unroll
            for (int e = 0; e < E; e++) {
                const u64 f0 = a0[e].reduce(M), f1 = a1[e].reduce(M);
                a0[e].clear(), a1[e].clear();
                a0[e].lo = f0, a1[e].lo = f1; // the folded sum counts as one more (tiny) term: 16 products + 2^60 < 2^124
            }
        }
    }
    if (m < ell) {
        u64 *ac = acc + (size_t)b * 2 * (ell + 1) * N;
        u64 *o0 = ac + ((size_t)0 * (ell + 1) + m) * N, *o1 = ac + ((size_t)1 * (ell + 1) + m) * N;
#pragma unroll
        for (int e = 0; e < E; e++) o0[g[e]] = a0[e].reduce(M), o1[g[e]] = a1[e].reduce(M);
    } else {
        const int p_lo = MERGE ? 0 : psel, p_hi = MERGE ? 1 : psel;
        for (int p = p_lo; p <= p_hi; p++) {
            u64 r[E];
#pragma unroll
            for (int e = 0; e < E; e++) r[e] = p == 0 ? a0[e].reduce(M) : a1[e].reduce(M);
            u64 *o = acc + (((size_t)b * 2 + p) * (ell + 1) + ell) * N;
            if (lds_used) __syncthreads();
            ntt_tile_x<K, LOGE, false, true, false, true, false>(r, M, itw + ((size_t)sp << logN), logN, blockIdx.x, nold,
                                                                 [=](int gi, u64 v) { o[gi] = v; }, lds);
            lds_used = true;
        }
    }
}

// opcode 10, third launch (source at 1 prime; with 2 the composition in the loader spills and the separate launches are
// faster): z = b*t + k.  The loader re-encodes coefficient g of item b from its decrypted
// coefficient-domain limbs (CRT compose, conjugate projection, scale, round), reduces it into target prime k, and the tile
// runs the first forward phase -- reencode_lift_batch_kernel + launch_ntt_cols_fwd without the round trip through ptx.
template <int K, int LOGE, int ELL>
__global__ __launch_bounds__(kTileThreads) void f_boot_reencode_fcols_kernel(const u64 *__restrict__ pt, u64 *__restrict__ ptx,
                                                                              const BootItem *__restrict__ items, int t,
                                                                              const DModulus *__restrict__ mods, const CrtDev crt,
                                                                              const u64 *__restrict__ tw, int logN)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    const int z = blockIdx.y, k = z % t, b = z / t;
    const size_t N = (size_t)1 << logN;
    const double ratio = items[b].ratio;
    const u64 *cf = pt + (size_t)b * ELL * N;
    u64 *out = ptx + (size_t)z * N;
    const DModulus M = mods[k];
    ntt_tile<K, LOGE, true, false, false>(
        M, tw + ((size_t)k << logN), logN, blockIdx.x,
        [=](int g) { return residue_of_double(reencoded_coeff_fixed<ELL>(cf, (size_t)g, N, mods, crt, ratio), M); },
        [=](int g, u64 v) { out[g] = v; }, lds);
}

// opcode 10, last launch: z = b*t + i.  dst.c0 = zenc.c0 + NTT(re-encoded plaintext), dst.c1 = zenc.c1
template <int K, int LOGE>
__global__ __launch_bounds__(kTileThreads) void f_frows_boot_final_kernel(const u64 *__restrict__ ptx, const BootItem *__restrict__ items,
                                                                           int t, const DModulus *__restrict__ mods,
                                                                           const u64 *__restrict__ tw, int logN)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    const int z = blockIdx.y, i = z % t;
    const size_t N = (size_t)1 << logN;
    const BootItem it = items[z / t];
    const DModulus M = mods[i];
    const u64 *in = ptx + (size_t)z * N;
    const u64 *z0 = it.zenc + (size_t)i * N, *z1 = it.zenc + ((size_t)t + i) * N;
    u64 *o0 = it.dst.limb(0, i, N), *o1 = it.dst.limb(1, i, N);
    ntt_tile<K, LOGE, false, false, true>(
        M, tw + ((size_t)i << logN), logN, blockIdx.x, [=](int g) { return in[g]; },
        [=](int g, u64 v) {
            o0[g] = addmod(v, z0[g], M.q);
            o1[g] = z1[g];
        },
        lds);
}

// ---- launchers (K and geometry dispatch: tile_dispatch.hpp) ---------------------------------------------------------------
template <class Src>
static void launch_irows(const Context &c, Src src, u64 *out, long out_stride, int count, hipStream_t s)
{
    DC_GEO_SWITCH(c.k2, count, DC_LAUNCH((f_irows_kernel<KK, LE, Src>), grid, dim3(kTileThreads), 0, s, src, out, out_stride,
                                                  c.d_mods, c.d_itw, c.logN));
}

void f_irows_strided(const Context &c, const u64 *base, long stride, int prime_base, int period, u64 *out, long out_stride, int count,
                     hipStream_t s)
{
    launch_irows(c, SrcStrided{ base, stride, prime_base, period }, out, out_stride, count, s);
}
// L1 of a rotation batch: inverse ROWS phase of galois(c1).  z = b*l + j.  An inverse ROWS tile starts with 2^LOGE CONSECUTIVE coefficients per
// thread, so the Galois gather is 16-byte loads of aligned pairs (galois_gather) instead of one 8-byte load per coefficient through a loader
// functor (the geometries with at least two coefficients per thread: all three)
template <int K, int LOGE>
__global__ __launch_bounds__(kTileThreads) void f_irows_rot_kernel(const KsItem *__restrict__ items, int ell, u64 *__restrict__ out,
                                                                    const DModulus *__restrict__ mods, const u64 *__restrict__ itw, int logN)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    constexpr int E = 1 << LOGE;
    const int z = blockIdx.y, j = z % ell;
    const size_t N = (size_t)1 << logN;
    const KsItem &it = items[z / ell];
    u64 *o = out + (size_t)z * N;
    u64 x[E];
    galois_gather<E>(x, it.src.limb(1, j, N), (u32)tile_gidx<K, LOGE, false>(num_passes<LOGE>(K) - 1, logN, blockIdx.x, 0), it.elt, logN);
    auto nold = [](int) -> u64 { return 0; };
    ntt_tile_x<K, LOGE, false, true, false, true, false>(x, mods[j], itw + ((size_t)j << logN), logN, blockIdx.x, nold,
                                                         [=](int gi, u64 v) { o[gi] = v; }, lds);
}

void f_irows_rot_c1(const Context &c, const KsItem *items, int ell, u64 *out, int B, hipStream_t s)
{
    DC_GEO_SWITCH(c.k2, B * ell, DC_LAUNCH((f_irows_rot_kernel<KK, LE>), grid, dim3(kTileThreads), 0, s, items, ell, out, c.d_mods, c.d_itw,
                                                    c.logN));
}
void f_irows_rs_last(const Context &c, const RsItem *items, const SumSrc *srcs, int l, u64 *out, int B, hipStream_t s)
{
    DC_GEO_SWITCH(c.k2, 2 * B, DC_LAUNCH((f_irows_rs_kernel<KK, LE>), grid, dim3(kTileThreads), 0, s, items, srcs, l, out, c.d_mods,
                                                  c.d_itw, c.logN));
}
void f_irows_rs_single(const Context &c, CtView src, int l, u64 *out, hipStream_t s)
{
    launch_irows(c, SrcRsLast1{ src, l }, out, (long)c.N, 2, s);
}
void f_irows_decrypt(const Context &c, CtView ct, const u64 *sk, int ell, u64 *out, hipStream_t s)
{
    launch_irows(c, SrcDecrypt{ ct, sk, c.d_mods }, out, (long)c.N, ell, s);
}

void f_irows_tensor_c2(const Context &c, const MulItem *items, int ell, u64 *out, int B, hipStream_t s)
{
    launch_irows(c, SrcTensorC2{ items, c.d_mods, ell }, out, (long)c.N, B * ell, s);
}
void f_irows_decrypt_items(const Context &c, const BootItem *items, const SumSrc *srcs, const u64 *sk, int ell, u64 *out, int B,
                           hipStream_t s)
{
    DC_GEO_SWITCH(c.k2, B * ell, DC_LAUNCH((f_irows_boot_kernel<KK, LE>), grid, dim3(kTileThreads), 0, s, items, srcs, sk, ell, out,
                                                    c.d_mods, c.d_itw, c.logN));
}
void f_boot_reencode_fcols(const Context &c, const u64 *pt, u64 *ptx, const BootItem *items, int B, int ell, int t, CrtDev crt,
                           hipStream_t s)
{
    if (ell == 1) {
        DC_GEO_SWITCH(c.k1, B * t, DC_LAUNCH((f_boot_reencode_fcols_kernel<KK, LE, 1>), grid, dim3(kTileThreads), 0, s, pt, ptx, items,
                                                      t, c.d_mods, crt, c.d_tw, c.logN));
    } else {
        fprintf(stderr, "[dacapo_amd] f_boot_reencode_fcols: source level %d not instantiated\n", ell);
        abort();
    }
}
void f_frows_boot_final(const Context &c, const u64 *ptx, const BootItem *items, int B, int t, hipStream_t s)
{
    DC_GEO_SWITCH(c.k2, B * t, DC_LAUNCH((f_frows_boot_final_kernel<KK, LE>), grid, dim3(kTileThreads), 0, s, ptx, items, t,
                                                  c.d_mods, c.d_tw, c.logN));
}

static long ks_merge_special_min_wgs()
{ // option ks_merge_special_min_wgs: launches of at least this many workgroups (more than the chip holds at once: throughput, not one
  // workgroup's latency, is what counts) let one row of workgroups serve both special-prime accumulators; a huge value = never
    return (long)option(OPT_KS_MERGE_SPECIAL_MIN_WGS);
}

void f_ks_frows_mac(const Context &c, int mode, const u64 *ext, const u64 *target, const KsItem *items, const u64 *shared_key, u64 *acc,
                    int B, int ell, hipStream_t s, bool fold_base)
{
    const u64 *pmod = (fold_base && mode == 0) ? c.d_pmod : nullptr;
#define DC_FMAC(LEV, MD)                                                                                                                  \
    {                                                                                                                                     \
        constexpr int LE = LEV;                                                                                                           \
        const long wgs = (long)(c.N >> TileGeo<LE>::LOG) * (ell + 2) * B;                                                                \
        const int merge = wgs >= ks_merge_special_min_wgs() ? 1 : 0;                                                                      \
        const int items_fast = (B > 1 && B <= 65535 && option(OPT_KS_ITEMS_FAST)) ? 1 : 0;                                               \
        const dim3 grid((unsigned)(c.N >> TileGeo<LE>::LOG), (unsigned)(items_fast ? B : ell + 2 - merge), (unsigned)(items_fast ? ell + 2 - merge : B)); \
        if (merge) {                                                                                                                      \
            DC_K_SWITCH(c.k2, DC_LAUNCH((f_ks_frows_mac_kernel<KK, LE, MD, true>), grid, dim3(kTileThreads), 0, s, ext, target, items,   \
                                                 shared_key, acc, ell, c.K, c.d_mods, c.d_tw, c.d_itw, c.logN, pmod, items_fast));       \
        } else {                                                                                                                          \
            DC_K_SWITCH(c.k2, DC_LAUNCH((f_ks_frows_mac_kernel<KK, LE, MD, false>), grid, dim3(kTileThreads), 0, s, ext, target, items,  \
                                                 shared_key, acc, ell, c.K, c.d_mods, c.d_tw, c.d_itw, c.logN, pmod, items_fast));       \
        }                                                                                                                                 \
    }
    // (the radix-8 geometry was measured for this kernel too: 8 coefficients x two 128-bit accumulators per thread cost more in
    // occupancy than the saved LDS exchange returns -- config 3: 520 us against 455 us; profiles/r02_experiments.txt)
    if (use_tiny_tiles(c.N, (long)(ell + 2) * B)) {
        if (mode == 0) DC_FMAC(1, 0) else DC_FMAC(1, 1)
    } else {
        if (mode == 0) DC_FMAC(2, 0) else DC_FMAC(2, 1)
    }
#undef DC_FMAC
}

static long ks_merge_lift_min_wgs()
{ // option ks_merge_lift_min_wgs: launches of L2 / L6 that still have at least this many workgroups AFTER merging share the inverse COLS
  // phase among a source limb's target moduli (a single key switch at 13 primes has 5408 fine workgroups or 416 merged ones: merged it
  // leaves CUs idle, 115 us against 93); a huge value = never
    return (long)option(OPT_KS_MERGE_LIFT_MIN_WGS);
}
// (the tile geometry is chosen from the unmerged limb count either way: the merged form is a throughput form of the same launch)
#define DC_MERGED_LAUNCH(limbs, sources, per_source, KERNEL, ...)                                                                          \
    {                                                                                                                                     \
        const bool tiny = use_tiny_tiles(c.N, (limbs)), small = !tiny && use_small_tiles(c.N, (limbs));                                  \
        const int le = tiny ? 1 : small ? 2 : 3;                                                                                          \
        const bool merge = (per_source) > 1 && (long)(c.N >> (le == 1 ? TileGeo<1>::LOG : le == 2 ? TileGeo<2>::LOG : TileGeo<3>::LOG)) * (sources) >= ks_merge_lift_min_wgs(); \
        if (!merge) {                                                                                                                     \
            DC_GEO_SWITCH(c.k1, (limbs), DC_LAUNCH((KERNEL<KK, LE, false>), grid, dim3(kTileThreads), 0, s, __VA_ARGS__));      \
        } else if (le == 1) {                                                                                                             \
            constexpr int LE = 1;                                                                                                         \
            const dim3 grid((unsigned)(c.N >> TileGeo<LE>::LOG), (unsigned)(sources));                                                    \
            DC_K_SWITCH(c.k1, DC_LAUNCH((KERNEL<KK, LE, true>), grid, dim3(kTileThreads), 0, s, __VA_ARGS__))                    \
        } else if (le == 2) {                                                                                                             \
            constexpr int LE = 2;                                                                                                         \
            const dim3 grid((unsigned)(c.N >> TileGeo<LE>::LOG), (unsigned)(sources));                                                    \
            DC_K_SWITCH(c.k1, DC_LAUNCH((KERNEL<KK, LE, true>), grid, dim3(kTileThreads), 0, s, __VA_ARGS__))                    \
        } else {                                                                                                                          \
            constexpr int LE = 3;                                                                                                         \
            const dim3 grid((unsigned)(c.N >> TileGeo<LE>::LOG), (unsigned)(sources));                                                    \
            DC_K_SWITCH(c.k1, DC_LAUNCH((KERNEL<KK, LE, true>), grid, dim3(kTileThreads), 0, s, __VA_ARGS__))                    \
        }                                                                                                                                 \
    }

void f_ks_icols_lift_fcols(const Context &c, const u64 *digits, u64 *ext, int B, int ell, hipStream_t s)
{
    DC_MERGED_LAUNCH(B * ell * ell, B * ell, ell, f_ks_icols_lift_fcols_kernel, digits, ext, ell, c.K - 1, c.d_mods, c.d_tw, c.d_itw, c.logN, c.twc2())
}

void f_dr_icols_lift_fcols(const Context &c, const u64 *last, long last_stride, u64 *tmp, int polys, int cnt, int l, hipStream_t s)
{
    DC_MERGED_LAUNCH(polys * cnt, polys, cnt, f_dr_icols_lift_fcols_kernel, last, last_stride, tmp, cnt, l, c.K, c.d_mods, c.d_half_mod, c.d_tw,
                     c.d_itw, c.logN, c.twc2())
}
#undef DC_MERGED_LAUNCH

void f_ks_lift_fcols(const Context &c, const u64 *digits, u64 *ext, int B, int ell, hipStream_t s)
{
    if ((c.k1 == 7 || c.k1 == 8) && use_wide_tiles(c.N, (long)B * ell * ell) && !use_small_tiles(c.N, (long)B * ell * ell)) { // round 5: radix-16 COLS tiles
        constexpr int LE = 4;                                                                                               // (ntt_tile.hpp; two phase sizes instantiated)
        const dim3 grid((unsigned)(c.N >> TileGeo<LE>::LOG), (unsigned)(B * ell * ell));
        if (c.k1 == 7)
            DC_LAUNCH((f_ks_lift_fcols_kernel<7, LE>), grid, dim3(kTileThreads), 0, s, digits, ext, ell, c.K - 1, c.d_mods, c.d_tw, c.logN, c.twc2());
        else
            DC_LAUNCH((f_ks_lift_fcols_kernel<8, LE>), grid, dim3(kTileThreads), 0, s, digits, ext, ell, c.K - 1, c.d_mods, c.d_tw, c.logN, c.twc2());
        return;
    }
    DC_GEO_SWITCH(c.k1, B * ell * ell, DC_LAUNCH((f_ks_lift_fcols_kernel<KK, LE>), grid, dim3(kTileThreads), 0, s, digits, ext, ell,
                                                          c.K - 1, c.d_mods, c.d_tw, c.logN, c.twc2()));
}

void f_dr_lift_fcols(const Context &c, const u64 *last, long last_stride, u64 *tmp, int polys, int cnt, int l, hipStream_t s)
{
    DC_GEO_SWITCH(c.k1, polys * cnt, DC_LAUNCH((f_dr_lift_fcols_kernel<KK, LE>), grid, dim3(kTileThreads), 0, s, last, last_stride,
                                                        tmp, cnt, l, c.K, c.d_mods, c.d_half_mod, c.d_tw, c.logN, c.twc2()));
}

void f_frows_final(const Context &c, int mode, const u64 *tmp, const void *items, const u64 *acc, int polys, int cnt, int l,
                   hipStream_t s, RsItem single, const u64 *plain, const SumSrc *srcs, const Handoff &h, bool base_folded)
{
    const int folded = base_folded ? 1 : 0;
    if (h.cont != CONT_NONE) {
        const int groups = h.cont == CONT_BOOT ? polys / 2 : polys; // CONT_BOOT: one workgroup per (item, limb) does both polynomials
#define DC_CONT(MD, CT)                                                                                                                \
    DC_GEO_SWITCH(c.k2, groups * cnt, DC_LAUNCH((f_frows_final_cont_kernel<KK, LE, MD, CT>), grid, dim3(kTileThreads), 0, s, tmp, items, \
                                                         srcs, acc, cnt, l, c.K, c.d_mods, c.d_inv_last, c.d_tw, c.d_itw, c.logN, h, folded))
        if (mode == 4 && h.cont == CONT_RS) {
            DC_CONT(4, CONT_RS);
        } else if (mode == 2 && h.cont == CONT_MUL) {
            DC_CONT(2, CONT_MUL);
        } else if (mode == 2 && h.cont == CONT_BOOT) {
            DC_CONT(2, CONT_BOOT);
        } else if (mode == 0 && h.cont == CONT_BOOT) {
            DC_CONT(0, CONT_BOOT);
        } else {
            fprintf(stderr, "[dacapo_amd] f_frows_final: no continuation kernel for mode %d -> consumer kind %d\n", mode, h.cont);
            abort();
        }
#undef DC_CONT
        return;
    }
#define DC_FINAL(MD)                                                                                                                   \
    DC_GEO_SWITCH(c.k2, polys * cnt, DC_LAUNCH((f_frows_final_kernel<KK, LE, MD>), grid, dim3(kTileThreads), 0, s, tmp, items, single, \
                                                        plain, srcs, acc, cnt, l, c.K, c.d_mods, c.d_inv_last, c.d_tw, c.logN, folded))
    switch (mode) {
    case 0: DC_FINAL(0); break;
    case 1: DC_FINAL(1); break;
    case 2: DC_FINAL(2); break;
    case 4: DC_FINAL(4); break;
    default: DC_FINAL(3); break;
    }
#undef DC_FINAL
}

} // namespace dacapo
