// EXTENSION -- not SEAL's scheme, not reached by the reference's SEAL runtime: hybrid key switching with GROUPED digits (dnum digits of
// `alpha` data primes, `ksp` special primes; Han-Ki 2020), what the reference's HEaaN runtime gets from its closed library for the
// N = 2^17 / 29-level configuration (HEAAN_HEVM.cpp:124-141 key generation, :386-399 bootstrap).  With SEAL's one prime per digit a
// key at N = 2^17 and 30 primes is 1.7 GB and a hop at the top level costs (l+1)(l+2) = 930 NTTs; with digits of 8 primes it is 0.3 GB
// and 4 (l + 8) + 16 + 2 l = 222.  The algorithm is oracle/ckks_oracle.c orc_keyswitch_hybrid, limb for limb (tests/test_gpu_hybrid.py);
// with ksp = alpha = 1 it is SEAL's switch_key_inplace and the fused launch sequences of fused_ks.hip stay in charge.
//
// Launch sequence of a batch of B key switches at level l (G = ceil(l / alpha) digits, M = l + ksp moduli, E = G M - l raised limbs):
//   prepare   rotation: dst.c0 = galois(src.c0), digits[slot] = src.c1 -- taken BEFORE the automorphism, so hops of one source share a
//             decomposition ("hoisting"), the Galois permutation is applied to the raised limbs in the NTT domain by `mac`'s loads
//             |   ct x ct: tensor product, target = digits = c2
//   iNTT      digits [B][l]                                   (batched transforms of ntt_kernels.hip / ntt_full.hip)
//   mod-up    ext[b][g][e] = sum_{i in S_g} [x_i qhat_i^-1]_{q_i} (Q_g / q_i)  mod m_e   -- the base conversion, a [|S_g| x (M - |S_g|)]
//             constant matrix applied to every coefficient: 128-bit lazy accumulators, one reduction per output
//   NTT       ext [B][E]
//   mac       acc[b][c][m] = sum_g operand_g[m] * key[g][c][m]   (operand = the NTT-form target limb itself when m is in S_g)
//   iNTT      the 2 ksp special-prime limbs of every item
//   mod-down  t_i = sum_j [(r_j + floor(P/2)) phat_j^-1]_{p_j} (P / p_j) - floor(P/2)  mod q_i
//   NTT       t [B][2][l]
//   final     dst.c += (acc_c - t_c) P^-1
// HBM-bound element-wise kernels around the transforms; 16-byte accesses, constants through scalar loads.
#include "plan.hpp"

namespace dacapo {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
constexpr int kHT = 256;
constexpr int kHybMaxAlpha = 16;

__device__ __forceinline__ u32 hyb_galois_src(u32 k, u32 elt, int logN)
{ // GaloisTool::apply_galois_ntt index map (poly_kernels.hip)
    const u32 r = (__brev(k) >> (32 - logN)) * 2u + 1u;
    const u32 idx = ((elt * r) >> 1) & ((1u << logN) - 1u);
    return __brev(idx) >> (32 - logN);
}

struct HybSingle { // one key switch by value (the one-instruction-at-a-time loop, the kernel-level C ABI): out = base + KS(target)
    CtView out;
    const u64 *base0 = nullptr, *base1 = nullptr, *key = nullptr;
};

// rotation items: dst.c0 = galois(src.c0); digits[slot] = src.c1 AS IT IS -- the digits are taken before the automorphism (see the header and
// oracle orc_rotate_ks_hybrid), so hops of one source ciphertext share one decomposition: every item of a slot writes the same limbs.
// `single` (items == nullptr): one hop by value.  grid = (N/512, l, B)
__global__ __launch_bounds__(kHT) void hyb_prepare_rot_kernel(const KsItem *__restrict__ items, KsItem single, u64 *__restrict__ digits, int ell,
                                                               size_t N, int logN, int use_slots)
{
    const int i = blockIdx.y, b = blockIdx.z;
    const KsItem it = items ? items[b] : single;
    const size_t k = ((size_t)blockIdx.x * kHT + threadIdx.x) * 2;
    const u32 g = hyb_galois_src((u32)k, it.elt, logN); // an aligned pair of outputs reads an aligned pair of inputs, possibly swapped
    const u64x2 v0 = *reinterpret_cast<const u64x2 *>(it.src.limb(0, i, N) + (g & ~1u));
    *reinterpret_cast<u64x2 *>(it.dst.limb(0, i, N) + k) = (g & 1u) ? u64x2{ v0.y, v0.x } : v0;
    const size_t slot = use_slots ? it.slot : (size_t)b;
    *reinterpret_cast<u64x2 *>(digits + (slot * ell + i) * N + k) = *reinterpret_cast<const u64x2 *>(it.src.limb(1, i, N) + k);
}

// ct x ct items: dst.c0 = a0 b0, dst.c1 = a0 b1 + a1 b0, target[b] = digits[b] = a1 b1.  grid = (N/512, l, B)
__global__ __launch_bounds__(kHT) void hyb_prepare_mul_kernel(const MulItem *__restrict__ items, u64 *__restrict__ target, u64 *__restrict__ digits,
                                                               int ell, size_t N, const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, b = blockIdx.z;
    const MulItem it = items[b];
    const DModulus m = mods[i];
    const size_t k = ((size_t)blockIdx.x * kHT + threadIdx.x) * 2;
    const u64x2 a0 = *reinterpret_cast<const u64x2 *>(it.a.limb(0, i, N) + k), a1 = *reinterpret_cast<const u64x2 *>(it.a.limb(1, i, N) + k);
    const u64x2 b0 = *reinterpret_cast<const u64x2 *>(it.b.limb(0, i, N) + k), b1 = *reinterpret_cast<const u64x2 *>(it.b.limb(1, i, N) + k);
    u64x2 c0, c1, c2;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        c0[e] = mulmod(a0[e], b0[e], m);
        Acc128 acc;
        acc.clear();
        acc.mac(a0[e], b1[e]);
        acc.mac(a1[e], b0[e]);
        c1[e] = acc.reduce(m);
        c2[e] = mulmod(a1[e], b1[e], m);
    }
    *reinterpret_cast<u64x2 *>(it.dst.limb(0, i, N) + k) = c0;
    *reinterpret_cast<u64x2 *>(it.dst.limb(1, i, N) + k) = c1;
    const size_t o = ((size_t)b * ell + i) * N + k;
    *reinterpret_cast<u64x2 *>(target + o) = c2;
    if (digits) *reinterpret_cast<u64x2 *>(digits + o) = c2; // (the fused sequence's first inverse phase reads `target` itself)
}

__global__ __launch_bounds__(kHT) void hyb_copy_kernel(u64 *__restrict__ dst, const u64 *__restrict__ src)
{
    const size_t k = ((size_t)blockIdx.x * kHT + threadIdx.x) * 2;
    *reinterpret_cast<u64x2 *>(dst + k) = *reinterpret_cast<const u64x2 *>(src + k);
}

// mod-up: digits [B][l][N] (coefficient domain) -> ext [B][E][N].  grid = (N/512, G, B)
__global__ __launch_bounds__(kHT) void hyb_modup_kernel(const u64 *__restrict__ digits, u64 *__restrict__ ext, int ell, int ksp, int alpha, int L,
                                                         int E, size_t N, const DModulus *__restrict__ mods, const u64 *__restrict__ up)
{
    const int g = blockIdx.y, b = blockIdx.z, M = ell + ksp;
    const int lo = g * alpha, hi = min(lo + alpha, ell), a = hi - lo;
    const size_t k = ((size_t)blockIdx.x * kHT + threadIdx.x) * 2;
    u64x2 y[kHybMaxAlpha];
#pragma unroll
    for (int t = 0; t < kHybMaxAlpha; t++) {
        if (t < a) {
            const DModulus mi = mods[lo + t];
            const u64x2 x = *reinterpret_cast<const u64x2 *>(digits + ((size_t)b * ell + lo + t) * N + k);
            const u64 inv = up[lo + t];
            y[t].x = mulmod(x.x, inv, mi), y[t].y = mulmod(x.y, inv, mi);
        }
    }
    u64 *out = ext + ((size_t)b * E + (size_t)g * (M - alpha)) * N + k; // every earlier digit is a full one
    const u64 *w = up + ell;
    for (int mi = 0; mi < M; mi++) {
        if (mi >= lo && mi < hi) continue;
        const DModulus m = mods[mi < ell ? mi : L + (mi - ell)];
        Acc128 a0, a1;
        a0.clear(), a1.clear();
#pragma unroll
        for (int t = 0; t < kHybMaxAlpha; t++) {
            if (t < a) {
                const u64 c = w[(size_t)(lo + t) * M + mi];
                // (mixed chains: a residue of a wider prime is reduced into a narrower target first, or the 128-bit sum leaves the range
                // 2^(2b+4) its reduction takes -- the generic-width build only; wave-uniform)
                a0.mac(fw_narrow(m.delta) ? canon(y[t].x, m) : y[t].x, c);
                a1.mac(fw_narrow(m.delta) ? canon(y[t].y, m) : y[t].y, c);
            }
        }
        u64x2 r;
        r.x = a0.reduce(m), r.y = a1.reduce(m);
        *reinterpret_cast<u64x2 *>(out) = r;
        out += N;
    }
}

// inner products with the key.  grid = (N/512, B, M).  MODE 0 rotation items (per-item key; every operand -- a raised limb of the item's
// slot, or the NTT-form limb of src.c1 itself where the modulus belongs to the digit -- is read THROUGH the item's Galois permutation),
// 1 ct x ct items (shared key), 2 one key switch by value
template <int MODE>
__global__ __launch_bounds__(kHT) void hyb_mac_kernel(u64 *__restrict__ accq, u64 *__restrict__ accp, const u64 *__restrict__ ext,
                                                       const u64 *__restrict__ target, const void *__restrict__ items, KsItem single,
                                                       const u64 *__restrict__ shared_key, int ell, int ksp, int alpha, int L, int K, int E, size_t N,
                                                       int logN, int use_slots, const DModulus *__restrict__ mods, const u64 *__restrict__ pmod)
{
    // (items fastest: the items of a hoisted batch read the same raised limbs -- through their own Galois maps -- within a few MiB of traffic
    // of each other instead of a whole pass over the key apart)
    const int mi = blockIdx.z, b = blockIdx.y, M = ell + ksp, pm = mi < ell ? mi : L + (mi - ell);
    const DModulus Md = mods[pm];
    KsItem it{};
    if (MODE == 0) it = items ? static_cast<const KsItem *>(items)[b] : single;
    const u64 *key = MODE == 0 ? it.key : shared_key;
    const size_t k = ((size_t)blockIdx.x * kHT + threadIdx.x) * 2;
    const size_t slot = MODE == 0 ? (use_slots ? it.slot : (size_t)b) : (size_t)b;
    u32 gsrc = (u32)k;
    if (MODE == 0) gsrc = hyb_galois_src((u32)k, it.elt, logN);
    const int G = (ell + alpha - 1) / alpha;
    Acc128 a0[2], a1[2];
#pragma unroll
    for (int e = 0; e < 2; e++) a0[e].clear(), a1[e].clear();
    // (the loads of digit g + 1 are issued before digit g's products: this kernel is bound by the key's bytes, and with one digit's three
    // 16-byte loads per thread in flight it ran at 0.65 of the HBM peak)
    auto fetch = [&](int g, u64x2 &x, u64x2 &y0, u64x2 &y1) {
        const int lo = g * alpha, hi = min(lo + alpha, ell);
        const bool own = mi >= lo && mi < hi;
        const u64 *op = own ? (MODE == 0 ? it.src.limb(1, mi, N) : target + ((size_t)b * ell + mi) * N)
                            : ext + (slot * E + (size_t)g * (M - alpha) + (mi < lo ? mi : mi - (hi - lo))) * N;
        x = *reinterpret_cast<const u64x2 *>(MODE == 0 ? op + (gsrc & ~1u) : op + k);
        y0 = *reinterpret_cast<const u64x2 *>(key + (((size_t)g * 2 + 0) * K + pm) * N + k);
        y1 = *reinterpret_cast<const u64x2 *>(key + (((size_t)g * 2 + 1) * K + pm) * N + k);
    };
    u64x2 xn, y0n, y1n;
    fetch(0, xn, y0n, y1n);
    for (int g = 0; g < G; g++) { // G <= 16: the 128-bit sums stay below 2^124
        u64x2 x = xn;
        const u64x2 y0 = y0n, y1 = y1n;
        if (g + 1 < G) fetch(g + 1, xn, y0n, y1n);
        if (MODE == 0 && (gsrc & 1u)) x = u64x2{ x.y, x.x };
#pragma unroll
        for (int e = 0; e < 2; e++) {
            a0[e].mac(x[e], y0[e]);
            a1[e].mac(x[e], y1[e]);
        }
    }
    u64x2 o0, o1;
#pragma unroll
    for (int e = 0; e < 2; e++) o0[e] = a0[e].reduce(Md), o1[e] = a1[e].reduce(Md);
    if (MODE == 0 && pmod && mi < ell) {
        // fused sequence: the rotation's base term rides on the accumulator -- galois(c0) + (acc - t) P^-1 = ((acc + P galois(c0)) - t) P^-1
        // exactly, so the last kernel has no gather to do and nobody materialises galois(c0) (this kernel has the Galois index in hand,
        // 16-byte accesses, and arithmetic to spare next to the key's bytes)
        const u64x2 v = *reinterpret_cast<const u64x2 *>(it.src.limb(0, mi, N) + (gsrc & ~1u));
        const u64 P = pmod[pm];
        o0.x = addmod(o0.x, mulmod((gsrc & 1u) ? v.y : v.x, P, Md), Md.q);
        o0.y = addmod(o0.y, mulmod((gsrc & 1u) ? v.x : v.y, P, Md), Md.q);
    }
    if (mi < ell) {
        *reinterpret_cast<u64x2 *>(accq + (((size_t)b * 2 + 0) * ell + mi) * N + k) = o0;
        *reinterpret_cast<u64x2 *>(accq + (((size_t)b * 2 + 1) * ell + mi) * N + k) = o1;
    } else {
        *reinterpret_cast<u64x2 *>(accp + (((size_t)b * 2 + 0) * ksp + (mi - ell)) * N + k) = o0;
        *reinterpret_cast<u64x2 *>(accp + (((size_t)b * 2 + 1) * ksp + (mi - ell)) * N + k) = o1;
    }
}

// lazy sums (option hyb_lazy_sum; hybrid_fused.hip hybf_rotate_sum): the inner products of ALL rotations of a group into ONE accumulator pair.
// grid = (N/512, groups, M); groups[g]: elt = first item, slot = item count.  Per item exactly hyb_mac_kernel<0>'s arithmetic (its own key,
// Galois map, decomposition slot and base term P galois(c0)), reduced to the canonical residue and added to the running sum -- so the sum equals
// the sum of the items' accumulators limb for limb, and no item's accumulator crosses HBM (2 (l + ksp) limbs written and read again per
// item otherwise: a third of this kernel's traffic at 12 primes, where a convolution's sums run).
__global__ __launch_bounds__(kHT) void hyb_mac_group_kernel(u64 *__restrict__ accq, u64 *__restrict__ accp, const u64 *__restrict__ ext,
                                                            const KsItem *__restrict__ items, const KsItem *__restrict__ groups, int ell, int ksp,
                                                            int alpha, int L, int K, int E, size_t N, int logN, int use_slots,
                                                            const DModulus *__restrict__ mods, const u64 *__restrict__ pmod)
{
    const int mi = blockIdx.z, gr = blockIdx.y, M = ell + ksp, pm = mi < ell ? mi : L + (mi - ell);
    const DModulus Md = mods[pm];
    const u32 first = groups[gr].elt, count = groups[gr].slot;
    const size_t k = ((size_t)blockIdx.x * kHT + threadIdx.x) * 2;
    const int G = (ell + alpha - 1) / alpha;
    const u64 P = mi < ell ? pmod[pm] : 0;
    u64x2 s0{ 0, 0 }, s1{ 0, 0 };
    for (u32 t = 0; t < count; t++) {
        const KsItem it = items[first + t];
        const size_t slot = use_slots ? it.slot : (size_t)(first + t);
        const u32 gsrc = hyb_galois_src((u32)k, it.elt, logN);
        Acc128 a0[2], a1[2];
#pragma unroll
        for (int e = 0; e < 2; e++) a0[e].clear(), a1[e].clear();
        auto fetch = [&](int g, u64x2 &x, u64x2 &y0, u64x2 &y1) {
            const int lo = g * alpha, hi = min(lo + alpha, ell);
            const bool own = mi >= lo && mi < hi;
            const u64 *op = own ? it.src.limb(1, mi, N) : ext + (slot * E + (size_t)g * (M - alpha) + (mi < lo ? mi : mi - (hi - lo))) * N;
            x = *reinterpret_cast<const u64x2 *>(op + (gsrc & ~1u));
            y0 = *reinterpret_cast<const u64x2 *>(it.key + (((size_t)g * 2 + 0) * K + pm) * N + k);
            y1 = *reinterpret_cast<const u64x2 *>(it.key + (((size_t)g * 2 + 1) * K + pm) * N + k);
        };
        u64x2 xn, y0n, y1n;
        fetch(0, xn, y0n, y1n);
        for (int g = 0; g < G; g++) {
            u64x2 x = xn;
            const u64x2 y0 = y0n, y1 = y1n;
            if (g + 1 < G) fetch(g + 1, xn, y0n, y1n);
            if (gsrc & 1u) x = u64x2{ x.y, x.x };
#pragma unroll
            for (int e = 0; e < 2; e++) {
                a0[e].mac(x[e], y0[e]);
                a1[e].mac(x[e], y1[e]);
            }
        }
        u64x2 o0, o1;
#pragma unroll
        for (int e = 0; e < 2; e++) o0[e] = a0[e].reduce(Md), o1[e] = a1[e].reduce(Md);
        if (mi < ell) { // the rotation's base term P galois(c0), as in hyb_mac_kernel<0>
            const u64x2 v = *reinterpret_cast<const u64x2 *>(it.src.limb(0, mi, N) + (gsrc & ~1u));
            o0.x = addmod(o0.x, mulmod((gsrc & 1u) ? v.y : v.x, P, Md), Md.q);
            o0.y = addmod(o0.y, mulmod((gsrc & 1u) ? v.x : v.y, P, Md), Md.q);
        }
        if (it.plain) { // double hoisting: pt * rot(x) enters the sum in the raised basis -- (P galois(c0) + <digits, key>) * pt limb by limb over
                        // Q and P, so that ModDown(sum) = sum_k pt_k galois(c0_k) + ModDown(sum_k pt_k <digits_k, key_k>): one division by P
            const u64x2 w = *reinterpret_cast<const u64x2 *>((mi < ell ? it.plain + (size_t)mi * N : it.plain_sp + (size_t)(mi - ell) * N) + k);
            o0.x = mulmod(o0.x, w.x, Md), o0.y = mulmod(o0.y, w.y, Md);
            o1.x = mulmod(o1.x, w.x, Md), o1.y = mulmod(o1.y, w.y, Md);
        }
        s0.x = addmod(s0.x, o0.x, Md.q), s0.y = addmod(s0.y, o0.y, Md.q);
        s1.x = addmod(s1.x, o1.x, Md.q), s1.y = addmod(s1.y, o1.y, Md.q);
    }
    if (mi < ell) {
        *reinterpret_cast<u64x2 *>(accq + (((size_t)gr * 2 + 0) * ell + mi) * N + k) = s0;
        *reinterpret_cast<u64x2 *>(accq + (((size_t)gr * 2 + 1) * ell + mi) * N + k) = s1;
    } else {
        *reinterpret_cast<u64x2 *>(accp + (((size_t)gr * 2 + 0) * ksp + (mi - ell)) * N + k) = s0;
        *reinterpret_cast<u64x2 *>(accp + (((size_t)gr * 2 + 1) * ksp + (mi - ell)) * N + k) = s1;
    }
}

// mod-down base conversion: accp [2B][ksp][N] (coefficient domain) -> tmp [2B][l][N].  grid = (N/512, 2B)
__global__ __launch_bounds__(kHT) void hyb_moddown_kernel(const u64 *__restrict__ accp, u64 *__restrict__ tmp, int ell, int ksp, int L, size_t N,
                                                           const DModulus *__restrict__ mods, const u64 *__restrict__ dn)
{
    const int z = blockIdx.y;
    const size_t k = ((size_t)blockIdx.x * kHT + threadIdx.x) * 2;
    const u64 *phat_inv = dn, *half_p = dn + ksp, *half_q = dn + 2 * ksp, *w = dn + 2 * ksp + 2 * L;
    u64x2 zz[kHybMaxAlpha];
#pragma unroll
    for (int j = 0; j < kHybMaxAlpha; j++) {
        if (j < ksp) {
            const DModulus mp = mods[L + j];
            const u64x2 r = *reinterpret_cast<const u64x2 *>(accp + ((size_t)z * ksp + j) * N + k);
            zz[j].x = mulmod(addmod(r.x, half_p[j], mp.q), phat_inv[j], mp);
            zz[j].y = mulmod(addmod(r.y, half_p[j], mp.q), phat_inv[j], mp);
        }
    }
    for (int i = 0; i < ell; i++) {
        const DModulus m = mods[i];
        Acc128 a0, a1;
        a0.clear(), a1.clear();
#pragma unroll
        for (int j = 0; j < kHybMaxAlpha; j++) {
            if (j < ksp) {
                const u64 c = w[(size_t)j * L + i];
                a0.mac(fw_narrow(m.delta) ? canon(zz[j].x, m) : zz[j].x, c);
                a1.mac(fw_narrow(m.delta) ? canon(zz[j].y, m) : zz[j].y, c);
            }
        }
        u64x2 r;
        r.x = submod(a0.reduce(m), half_q[i], m.q), r.y = submod(a1.reduce(m), half_q[i], m.q);
        *reinterpret_cast<u64x2 *>(tmp + ((size_t)z * ell + i) * N + k) = r;
    }
}

// dst.c (+)= (acc_c - t_c) P^-1.  grid = (N/512, l, 2B)
template <int MODE>
__global__ __launch_bounds__(kHT) void hyb_final_kernel(const u64 *__restrict__ accq, const u64 *__restrict__ tmp, const void *__restrict__ items,
                                                         HybSingle single, KsItem rot_single, int ell, int ksp, int L, size_t N,
                                                         const DModulus *__restrict__ mods, const u64 *__restrict__ dn)
{
    const int i = blockIdx.y, z = blockIdx.z, b = z >> 1, c = z & 1;
    const DModulus m = mods[i];
    const u64 pinv = dn[2 * ksp + L + i];
    const size_t k = ((size_t)blockIdx.x * kHT + threadIdx.x) * 2;
    const u64x2 a = *reinterpret_cast<const u64x2 *>(accq + ((size_t)z * ell + i) * N + k);
    const u64x2 t = *reinterpret_cast<const u64x2 *>(tmp + ((size_t)z * ell + i) * N + k);
    u64x2 v;
    v.x = mulmod(submod(a.x, t.x, m.q), pinv, m), v.y = mulmod(submod(a.y, t.y, m.q), pinv, m);
    u64 *dst;
    const u64 *base;
    if (MODE == 0) {
        const KsItem it = items ? static_cast<const KsItem *>(items)[b] : rot_single;
        dst = it.dst.limb(c, i, N) + k, base = c == 0 ? dst : nullptr; // c0 holds the permuted source c0 (prepare), c1 starts at zero
    } else if (MODE == 1) {
        const MulItem it = static_cast<const MulItem *>(items)[b];
        dst = it.dst.limb(c, i, N) + k, base = dst;                     // the tensor product's c0 / c1
    } else {
        dst = single.out.limb(c, i, N) + k;
        const u64 *bp = c == 0 ? single.base0 : single.base1;
        base = bp ? bp + (size_t)i * N + k : nullptr;
    }
    if (base) {
        const u64x2 o = *reinterpret_cast<const u64x2 *>(base);
        v.x = addmod(v.x, o.x, m.q), v.y = addmod(v.y, o.y, m.q);
    }
    *reinterpret_cast<u64x2 *>(dst) = v;
}

// ---- the two base conversions on the matrix cores ------------------------------------------------------------------------------
// out[e][n] = sum_t y_t[n] w[t][e] mod m_e for 16 coefficients x 16 output moduli per MFMA tile (see context.hip for the operand
// encoding: balanced base-256 digits of y on the A side, of w 2^(8p) mod m on the B side; 8 accumulator planes r recombined as
// sum_r C_r 2^(8r), a signed 80-bit integer, then reduced).  One wave = a strip of 64 coefficients (4 A fragments kept in registers),
// one workgroup = 4 waves; the B fragments of a block of 16 moduli (8 x 16 bytes per lane) are loaded once per strip.
// DOWN = false: mod-up of digit g = blockIdx.y of item blockIdx.z; DOWN = true: mod-down of polynomial blockIdx.z (= 2 b + c).
typedef int v4i __attribute__((ext_vector_type(4)));
// (128 coefficients per wave = 8 A fragments kept 132 VGPRs alive: three waves per SIMD; with 64 it is 116 and four, and twice the waves to
// hide the strip's load -> MFMA -> store chain: hop at 7 / 14 primes 133 -> 127 / 199 -> 193 us, profiles/r04_experiments.txt)
constexpr int kConvStrip = 64; // coefficients per wave

// Recombination of the 8 accumulator planes, round 4 (round 3's form spent ~50 vector instructions per output on a signed 128-bit sum, and the
// matrix pipes sat idle 93 % of the launch).  The accumulators START at 2^20 (the MFMA's C operand), so every plane value c'_r = C_r + 2^20 is
// in [0, 2^21] and T' = sum_r c'_r 2^(8r) < 2^78 is an unsigned sum: two chains of three v_mad_u64_u32 (the shifts are multiplications by
// constants), one 96-bit assembly, ONE fold (T' >> 60 < 2^18, so (T' >> 60) d + (T' mod 2^60) < 2q), one conditional subtraction -- and the
// constant K0 = 2^20 (2^64 - 1) / 255 that the offsets added leaves with the caller's own final subtraction (k0 = K0 mod m, plus the
// mod-down's floor(P/2)).  ~23 instructions.  The generic-width build keeps the 128-bit reduction (T' >> b can exceed 32 bits there).
// (round 5: a conversion with 9..16 inputs -- the mod-down under 9 special primes -- runs TWO K-chunks per tile, 128 byte products per plane:
// |sum| <= 2^21, so its accumulators start at 2^21 and every plane value is in [0, 2^22]; T' < 2^78.01, T' >> 60 < 2^19: same recombination)
constexpr int kPlaneBias = 1 << 20;
// Round 5: the constant the plane biases add is no longer subtracted per output with two modular steps.  With e = -(K0 + post) mod m (K0 = the
// biases' sum, post = the mod-down's floor(P/2) mod q_i; one value per output modulus, computed once per block of moduli),
// T' + e = sum_r c'_r 2^(8r) + e is congruent to the wanted value itself, and the output is ONE fold of it -- below 2q, which is all its only
// consumer (the first stage of a forward transform, an F stage) asks for.  The conditional subtraction and the modular subtraction of k0
// (9-10 of the ~25 instructions per output) become one 64-bit add with carry.  (Putting e's base-256 digits into the MFMA's C operand
// instead was built first: 8 x 4 more live registers, 152 -> 212 per thread, two waves per SIMD instead of three.)
__device__ __forceinline__ u64 hyb_recombine(const v4i (&c)[8], int j, const DModulus &M, u64 e)
{
    const u64 slo = mad32((u32)c[3][j], 1u << 24, mad32((u32)c[2][j], 1u << 16, mad32((u32)c[1][j], 1u << 8, (u64)(u32)c[0][j])));
    const u64 shi = mad32((u32)c[7][j], 1u << 24, mad32((u32)c[6][j], 1u << 16, mad32((u32)c[5][j], 1u << 8, (u64)(u32)c[4][j])));
    const u64 l0 = slo + (shi << 32);
    const u64 lo = l0 + e; // (e < 2^60: T' + e < 2^79)
    const u64 hi = (shi >> 32) + (l0 < slo ? 1u : 0u) + (lo < l0 ? 1u : 0u);
#if DC_GENERIC_WIDTH
    return reduce128_any(hi, lo, M);
#else
    const u32 top = (u32)((hi << 4) | (lo >> 60));
    return mad32(top, M.delta, lo & ((1ull << 60) - 1)); // < 2^60 + 2^19 d < 2q: a lazy residue
#endif
}
// K0 mod m, K0 = 2^(20 + extra) * 0x0101010101010101 (what the plain plane biases add to every output)
__device__ __forceinline__ u64 hyb_plane_bias_mod(const DModulus &M, int extra = 0)
{
    const u64 ones = 0x0101010101010101ull;
    return reduce128_any(ones >> (44 - extra), ones << (20 + extra), M);
}

// PRE: the inputs already carry the conversion's per-input constant -- and, for DOWN, the rounding offset floor(P/2) phat_inv_j -- (the fused
// sequence folds them into the inverse transform's last stage and store, hybrid_fused.hip)
// CH: K-chunks per tile.  1 everywhere but the mod-down of a context with 9..12 special primes (config 4's 4-digit key shape: digits of 8
// under 9 special primes), which takes inputs 8..11 through a second, 32-byte-deep MFMA (v_mfma_i32_16x16x32_i8) into the same accumulators.
// (First version: a second 64-deep MFMA with its operands kept in registers -- 200 registers per thread, two waves per SIMD, 239 us per
// launch against the 8-input kernel's 146; the short tail with its B operand loaded where it is used: 162 registers, three waves.)
template <bool DOWN, bool PRE, int CH = 1>
__global__ __launch_bounds__(kHT) void hyb_conv_mfma_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, int ell, int ksp, int alpha, int L, int E,
                                                             size_t N, const DModulus *__restrict__ mods, const u64 *__restrict__ cst,
                                                             const v4i *__restrict__ btab, int nblk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kb = lane >> 4;
    const size_t n0 = ((size_t)blockIdx.x * 4 + wave) * kConvStrip;
    int a, lo, in_prime0, n_out;
    const u64 *inp;
    u64 *outp;
    const v4i *bt;
    if (!DOWN) {
        const int g = blockIdx.y, b = blockIdx.z, M = ell + ksp;
        lo = g * alpha;
        a = min(lo + alpha, ell) - lo;
        in_prime0 = lo;
        n_out = M - a;
        inp = in + ((size_t)b * ell + lo) * N;
        outp = out + ((size_t)b * E + (size_t)g * (M - alpha)) * N;
        bt = btab + (size_t)g * nblk * 8 * 64;
    } else {
        const int z = blockIdx.z;
        lo = 0, a = ksp, in_prime0 = L, n_out = ell;
        inp = in + (size_t)z * ksp * N;
        outp = out + (size_t)z * ell * N;
        bt = btab;
    }
    // A fragments: this lane's two inputs t = 8 ch + 2 kb, + 1 of its row, times the conversion's per-input constant, as balanced bytes
    static_assert(CH == 1 || DOWN, "only the mod-down has more than 8 inputs");
    constexpr int kBias = kPlaneBias << (CH - 1);
    v4i A[kConvStrip / 16];
    long A2[CH > 1 ? kConvStrip / 16 : 1]; // the tail chunk: input 8 + kb of this lane's row, 8 balanced bytes
    const u64 C8 = 0x8080808080808080ull;
    {
        u64 mul[2], add[2];
        DModulus mi[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int t = 2 * kb + h, tt = t < a ? t : 0;
            mi[h] = mods[in_prime0 + tt];
            mul[h] = DOWN ? cst[tt] : cst[lo + tt];          // phat_inv[j] | qhat_inv[i]
            add[h] = DOWN ? cst[ksp + tt] : 0;               // floor(P/2) mod p_j | -
        }
#pragma unroll
        for (int tile = 0; tile < kConvStrip / 16; tile++) {
            const size_t n = n0 + (size_t)tile * 16 + col;
            u64 y[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int t = 2 * kb + h;
                if (t < a && t < 8) {
                    u64 x = inp[(size_t)t * N + n];
                    if (DOWN && !PRE) x = addmod(x, add[h], mi[h].q);
                    if (!PRE) x = mulmod(x, mul[h], mi[h]);
                    y[h] = (x + C8) ^ C8;
                } else
                    y[h] = 0;
            }
            A[tile] = v4i{ (int)(u32)y[0], (int)(u32)(y[0] >> 32), (int)(u32)y[1], (int)(u32)(y[1] >> 32) };
        }
    }
    if constexpr (CH > 1) {
        const int t = 8 + kb, tt = t < a ? t : 0;
        const DModulus mt = mods[in_prime0 + tt];
        const u64 mulx = cst[tt], addx = cst[ksp + tt];
#pragma unroll
        for (int tile = 0; tile < kConvStrip / 16; tile++) {
            u64 y = 0;
            if (t < a) {
                u64 x = inp[(size_t)t * N + n0 + (size_t)tile * 16 + col];
                if (!PRE) x = mulmod(addmod(x, addx, mt.q), mulx, mt);
                y = (x + C8) ^ C8;
            }
            A2[tile] = (long)y;
        }
    }
    const long *bt2 = reinterpret_cast<const long *>(bt + (size_t)nblk * 8 * 64); // (the tail's B fragments follow the level's main table)
    for (int blk = 0; blk < nblk; blk++) {
        if (blk * 16 >= n_out) break;
        v4i Bf[8];
#pragma unroll
        for (int r = 0; r < 8; r++) Bf[r] = bt[((size_t)blk * 8 + r) * 64 + lane];
        const int e = blk * 16 + col;
        const bool valid = e < n_out;
        int pm;
        u64 post = 0;
        if (!DOWN) {
            const int mi2 = e < lo ? e : e + a;
            pm = mi2 < ell ? mi2 : L + (mi2 - ell);
        } else {
            pm = e;
            post = valid ? cst[2 * ksp + e] : 0; // floor(P/2) mod q_i
        }
        const DModulus Mo = mods[valid ? pm : 0];
        // this lane's column = one output modulus: the constant hyb_recombine adds, once per block of 16 moduli
        const u64 k0 = addmod(hyb_plane_bias_mod(Mo, CH - 1), post, Mo.q);
        const u64 eb = k0 ? Mo.q - k0 : 0; // -(K0 + post) mod m, < 2^60
#pragma unroll
        for (int tile = 0; tile < kConvStrip / 16; tile++) {
            v4i c[8];
            size_t t2 = (size_t)blk * 8 * 64 + lane;
            if constexpr (CH > 1) asm volatile("" : "+v"(t2)); // (a per-tile value as far as the optimiser can tell: the tail's B operand is loaded
                                                               // here, out of L1, instead of living in 16 registers across the tiles)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                c[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[tile], Bf[r], v4i{ kBias, kBias, kBias, kBias }, 0, 0, 0);
                if constexpr (CH > 1) c[r] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A2[tile], bt2[t2 + (size_t)r * 64], c[r], 0, 0, 0);
            }
            if (valid) { // this lane: rows 4 kb .. 4 kb + 3 of the tile, column `col`
                u64 v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = hyb_recombine(c, j, Mo, eb);
                u64 *o = outp + (size_t)e * N + n0 + (size_t)tile * 16 + 4 * kb;
                *reinterpret_cast<u64x2 *>(o) = u64x2{ v[0], v[1] };
                *reinterpret_cast<u64x2 *>(o + 2) = u64x2{ v[2], v[3] };
            }
        }
    }
}

// the two base conversions (matrix cores when the context has their operand tables and the level is high enough, else the vector kernels).
// `prescaled`: inputs already multiplied by the per-input constant (fused sequence).  count = decompositions (up) / polynomials (down).
// OUTPUT RANGE (contract): the matrix-core form of the 60-bit build writes LAZY residues -- congruent to the converted value, one fold, below
// 2q, NOT canonical (hyb_recombine); the vector kernels and the generic-width build write canonical ones.  Every consumer must therefore
// fold its input before anything that needs x < q: a forward transform's first stage does (launch_ntt / launch_ntt_cols_fwd fold at stage 0,
// with and without twiddle pairs; ntt_full's pair pass takes x < 2q).  ntt_tile_core<PAIRS> without FOLD0, submod-based epilogues and
// anything that compares limbs (tests reading w.ext / w.tmp) are NOT valid consumers.  tests/test_gpu_hybrid.py
// test_lazy_conversion_outputs_feed_every_forward_transform runs each forward path behind it against the oracle.
void hyb_launch_conv(Context &c, bool down, bool prescaled, const u64 *in, u64 *out, int count, int ell, hipStream_t s)
{
    const size_t N = c.N;
    const int ksp = c.ksp, alpha = c.alpha, L = c.max_level(), G = c.hyb_groups(ell), E = c.hyb_ext(ell);
    // (below 4 primes a conversion has at most a 3 x 11 matrix: the vector kernels are faster there -- 115 vs 127 us per hop at level 1, N = 2^17)
    const bool mfma = c.hyb_mfma && N >= 4 * kConvStrip && ell >= 4;
    const unsigned gc = (unsigned)(N / (4 * kConvStrip)), gx = (unsigned)(N / (2 * kHT));
    if (prescaled && !mfma) {
        fprintf(stderr, "[dacapo_amd] hyb_launch_conv: pre-scaled inputs need the matrix-core form\n");
        abort();
    }
    const v4i *bup = reinterpret_cast<const v4i *>(c.d_hyb_bup + (mfma ? c.hyb_bup_off[(size_t)ell] : 0));
    const v4i *bdn = reinterpret_cast<const v4i *>(c.d_hyb_bdn + (mfma ? c.hyb_bdn_off[(size_t)ell] : 0));
    if (!down) {
        if (mfma && prescaled)
            DC_LAUNCH((hyb_conv_mfma_kernel<false, true>), dim3(gc, (unsigned)G, (unsigned)count), dim3(kHT), 0, s, in, out, ell, ksp, alpha, L, E,
                               N, c.d_mods, c.hyb_up(ell), bup, c.hyb_up_blocks(ell));
        else if (mfma)
            DC_LAUNCH((hyb_conv_mfma_kernel<false, false>), dim3(gc, (unsigned)G, (unsigned)count), dim3(kHT), 0, s, in, out, ell, ksp, alpha, L, E,
                               N, c.d_mods, c.hyb_up(ell), bup, c.hyb_up_blocks(ell));
        else
            DC_LAUNCH(hyb_modup_kernel, dim3(gx, (unsigned)G, (unsigned)count), dim3(kHT), 0, s, in, out, ell, ksp, alpha, L, E, N, c.d_mods,
                               c.hyb_up(ell));
    } else {
        if (mfma && prescaled && ksp > 8)
            DC_LAUNCH((hyb_conv_mfma_kernel<true, true, 2>), dim3(gc, 1, (unsigned)count), dim3(kHT), 0, s, in, out, ell, ksp, alpha, L, E, N,
                               c.d_mods, c.d_hyb_dn, bdn, c.hyb_dn_blocks(ell));
        else if (mfma && ksp > 8)
            DC_LAUNCH((hyb_conv_mfma_kernel<true, false, 2>), dim3(gc, 1, (unsigned)count), dim3(kHT), 0, s, in, out, ell, ksp, alpha, L, E, N,
                               c.d_mods, c.d_hyb_dn, bdn, c.hyb_dn_blocks(ell));
        else if (mfma && prescaled)
            DC_LAUNCH((hyb_conv_mfma_kernel<true, true>), dim3(gc, 1, (unsigned)count), dim3(kHT), 0, s, in, out, ell, ksp, alpha, L, E, N,
                               c.d_mods, c.d_hyb_dn, bdn, c.hyb_dn_blocks(ell));
        else if (mfma)
            DC_LAUNCH((hyb_conv_mfma_kernel<true, false>), dim3(gc, 1, (unsigned)count), dim3(kHT), 0, s, in, out, ell, ksp, alpha, L, E, N,
                               c.d_mods, c.d_hyb_dn, bdn, c.hyb_dn_blocks(ell));
        else
            DC_LAUNCH(hyb_moddown_kernel, dim3(gx, (unsigned)count), dim3(kHT), 0, s, in, out, ell, ksp, L, N, c.d_mods, c.d_hyb_dn);
    }
}

// lazy sums: G groups of rotation items -> accq = w.acc [2G][ell][N], accp behind it [2G][ksp][N]
void hyb_launch_mac_groups(Context &c, const BatchWs &w, const KsItem *items, const KsItem *groups, int G, int use_slots, int ell, hipStream_t s)
{
    const size_t N = c.N;
    const int ksp = c.ksp, M = ell + ksp;
    DC_LAUNCH(hyb_mac_group_kernel, dim3((unsigned)(N / (2 * kHT)), (unsigned)G, (unsigned)M), dim3(kHT), 0, s, w.acc,
              w.acc + (size_t)G * 2 * ell * N, w.ext, items, groups, ell, ksp, c.alpha, c.max_level(), c.K, c.hyb_ext(ell), N, c.logN, use_slots, c.d_mods,
              c.d_pmod);
}

// inner products with the key: w.ext [U][E][N] (NTT form), w.target (modes 1, 2) -> accq = w.acc [2B][ell][N], accp behind it [2B][ksp][N]
void hyb_launch_mac(Context &c, int mode, const BatchWs &w, const void *items, KsItem rot_single, const u64 *key, int B, int use_slots, int ell,
                    hipStream_t s, bool fold_base)
{
    const size_t N = c.N;
    const int ksp = c.ksp, alpha = c.alpha, L = c.max_level(), K = c.K, M = ell + ksp, E = c.hyb_ext(ell);
    const dim3 grid((unsigned)(N / (2 * kHT)), (unsigned)B, (unsigned)M);
    u64 *accq = w.acc, *accp = w.acc + (size_t)B * 2 * ell * N;
#define DC_MAC(MD)                                                                                                                        \
    DC_LAUNCH(hyb_mac_kernel<MD>, grid, dim3(kHT), 0, s, accq, accp, w.ext, w.target, items, rot_single, key, ell, ksp, alpha, L, K, E, N, \
                       c.logN, use_slots, c.d_mods, fold_base ? c.d_pmod : (const u64 *)nullptr)
    if (mode == 0)
        DC_MAC(0);
    else if (mode == 1)
        DC_MAC(1);
    else
        DC_MAC(2);
#undef DC_MAC
}

// everything after `prepare`.  U = decompositions to compute (digits [U][l][N] in NTT form on entry, transformed in place here): U = B
// except for rotation batches whose items share sources (use_slots); MODE 1 / 2: target [B][l][N] NTT form.
template <int MODE>
static void hyb_core(Context &c, const BatchWs &w, const void *items, const u64 *shared_key, HybSingle single, KsItem rot_single, int B, int U,
                     int use_slots, int ell, hipStream_t s)
{
    const size_t N = c.N;
    const int ksp = c.ksp, L = c.max_level(), E = c.hyb_ext(ell);
    const unsigned gx = (unsigned)(N / (2 * kHT));
    u64 *accq = w.acc, *accp = w.acc + (size_t)B * 2 * ell * N;
    launch_ntt(c, true, w.digits, (long)N, U * ell, nullptr, 0, ell, s);
    hyb_launch_conv(c, false, false, w.digits, w.ext, U, ell, s);
    launch_ntt(c, false, w.ext, (long)N, U * E, c.hyb_pidx(ell), 0, E, s);
    hyb_launch_mac(c, MODE, w, items, rot_single, MODE == 2 ? single.key : shared_key, B, use_slots, ell, s, false);
    launch_ntt(c, true, accp, (long)N, 2 * B * ksp, nullptr, L, ksp, s);
    hyb_launch_conv(c, true, false, accp, w.tmp, 2 * B, ell, s);
    launch_ntt(c, false, w.tmp, (long)N, 2 * B * ell, nullptr, 0, ell, s);
    DC_LAUNCH(hyb_final_kernel<MODE>, dim3(gx, (unsigned)ell, (unsigned)(2 * B)), dim3(kHT), 0, s, accq, w.tmp, items, single, rot_single, ell,
                       ksp, L, N, c.d_mods, c.d_hyb_dn);
}

// the fused sequence (hybrid_fused.hip; option hyb_fuse != 0, the default)
void hybf_rotate_hops(Context &c, const BatchWs &w, const KsItem *d_items, int B, int ell, hipStream_t s, int unique);
void hybf_rotate_hop_single(Context &c, const Workspace &w, CtView dst, CtView src, u32 galois_elt, const u64 *galois_key, int ell, hipStream_t s);
void hybf_mul_relin_tail(Context &c, const BatchWs &w, const MulItem *d_items, const u64 *relin_key, int B, int ell, hipStream_t s);
void hybf_keyswitch(Context &c, const Workspace &w, CtView out, const u64 *base0, const u64 *base1, const u64 *target, const u64 *key, int ell,
                    hipStream_t s);
void hybf_rotate_sum(Context &c, const BatchWs &w, const KsItem *d_items, int B, const KsItem *d_groups, int G, int ell, hipStream_t s, int unique);
static bool hyb_fused() { return option(OPT_HYB_FUSE) != 0; }
bool hyb_lazy_sum_supported(const Context &c) { return c.hybrid() && hyb_fused() && c.N >= 512; }

void hyb_rotate_sum(Context &c, const BatchWs &w, const KsItem *d_items, int B, const KsItem *d_groups, int G, int ell, hipStream_t s, int unique)
{
    if (!hyb_lazy_sum_supported(c)) {
        fprintf(stderr, "[dacapo_amd] lazy sums need the fused grouped-digit sequence (option hyb_fuse != 0)\n");
        abort();
    }
    hybf_rotate_sum(c, w, d_items, B, d_groups, G, ell, s, unique);
}

void hyb_rotate_hops(Context &c, const BatchWs &w, const KsItem *d_items, int B, int ell, hipStream_t s, int unique)
{
    if (hyb_fused()) return hybf_rotate_hops(c, w, d_items, B, ell, s, unique);
    const int use_slots = unique > 0 ? 1 : 0, U = use_slots ? unique : B;
    DC_LAUNCH(hyb_prepare_rot_kernel, dim3((unsigned)(c.N / (2 * kHT)), (unsigned)ell, (unsigned)B), dim3(kHT), 0, s, d_items, KsItem{},
                       w.digits, ell, c.N, c.logN, use_slots);
    hyb_core<0>(c, w, d_items, nullptr, HybSingle{}, KsItem{}, B, U, use_slots, ell, s);
}

void hyb_rotate_hop_single(Context &c, const Workspace &w, CtView dst, CtView src, u32 galois_elt, const u64 *galois_key, int ell, hipStream_t s)
{
    if (hyb_fused()) return hybf_rotate_hop_single(c, w, dst, src, galois_elt, galois_key, ell, s);
    const KsItem it{ src, dst, galois_key, galois_elt, 0 };
    BatchWs bw{ nullptr, w.ks_digits, w.ks_ext, w.ks_acc, w.ks_tmp };
    DC_LAUNCH(hyb_prepare_rot_kernel, dim3((unsigned)(c.N / (2 * kHT)), (unsigned)ell, 1), dim3(kHT), 0, s, (const KsItem *)nullptr, it,
                       w.ks_digits, ell, c.N, c.logN, 0);
    hyb_core<0>(c, bw, nullptr, nullptr, HybSingle{}, it, 1, 1, 0, ell, s);
}

void hyb_mul_relin(Context &c, const BatchWs &w, const MulItem *d_items, const u64 *relin_key, int B, int ell, hipStream_t s)
{
    DC_LAUNCH(hyb_prepare_mul_kernel, dim3((unsigned)(c.N / (2 * kHT)), (unsigned)ell, (unsigned)B), dim3(kHT), 0, s, d_items, w.target,
                       hyb_fused() ? (u64 *)nullptr : w.digits, ell, c.N, c.d_mods);
    if (hyb_fused()) return hybf_mul_relin_tail(c, w, d_items, relin_key, B, ell, s);
    hyb_core<1>(c, w, d_items, relin_key, HybSingle{}, KsItem{}, B, B, 0, ell, s);
}

// Evaluator::switch_key_inplace's role for one ciphertext: out = (base0, base1) + KS(target).  target [l][N] NTT form, preserved.
void hyb_keyswitch(Context &c, const Workspace &w, CtView out, const u64 *base0, const u64 *base1, const u64 *target, const u64 *key, int ell,
                   hipStream_t s)
{
    if (hyb_fused()) return hybf_keyswitch(c, w, out, base0, base1, target, key, ell, s);
    const size_t N = c.N;
    // the batch scratch of one item: target is read in place, digits / ext / acc / tmp are the workspace's
    BatchWs bw{ const_cast<u64 *>(target), w.ks_digits, w.ks_ext, w.ks_acc, w.ks_tmp };
    DC_LAUNCH(hyb_copy_kernel, dim3((unsigned)((size_t)ell * N / (2 * kHT))), dim3(kHT), 0, s, w.ks_digits, target);
    hyb_core<2>(c, bw, nullptr, nullptr, HybSingle{ out, base0, base1, key }, KsItem{}, 1, 1, 0, ell, s);
}

} // namespace dacapo
