// Batched forms of the composite evaluator ops (plan.hpp): B independent ciphertext ops of one kind and level per
// launch sequence.  Same arithmetic, same order of modular operations per limb as ckks_ops.hip -- only the grid grows
// by the batch dimension and operands come from a device table of views.  Algorithms: SEAL 4.0
// Evaluator::switch_key_inplace / rescale_to_next / multiply [SEAL-upstream], reached from SEAL_HEVM.cpp:273,283,315-316.
#include <stdlib.h>

#include "plan.hpp"

namespace dacapo {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
constexpr int kBT = 256;

__device__ __forceinline__ u32 galois_src(u32 k, u32 elt, int logN)
{ // GaloisTool::apply_galois_ntt index map (see poly_kernels.hip)
    const u32 r = (__brev(k) >> (32 - logN)) * 2u + 1u;
    const u32 idx = ((elt * r) >> 1) & ((1u << logN) - 1u);
    return __brev(idx) >> (32 - logN);
}

// ckks_multiply prologue: dst.c0 = a0 b0, dst.c1 = a0 b1 + a1 b0, target[b] = a1 b1.  grid = (N/512, l, B)
__global__ __launch_bounds__(kBT) void b_tensor_kernel(const MulItem *__restrict__ items, u64 *__restrict__ target, int ell,
                                                        size_t N, const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, b = blockIdx.z;
    const MulItem it = items[b];
    const DModulus m = mods[i];
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
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
}

// inner products with the key of item b.  grid = (N/512, l+1 [m], B)
// MODE 0 (rotation): the j == m operand is the item's c1 read through its Galois permutation, key per item;
// MODE 1 (relinearisation): the j == m operand is target[b] (the c2 of the tensor product), one shared key.
template <int MODE>
__global__ __launch_bounds__(kBT) void b_ks_mac_kernel(u64 *__restrict__ acc, const u64 *__restrict__ ext,
                                                        const u64 *__restrict__ target, const KsItem *__restrict__ items,
                                                        const u64 *__restrict__ shared_key, int ell, int K, size_t N, int logN,
                                                        const DModulus *__restrict__ mods)
{
    const int m = blockIdx.y, b = blockIdx.z, sp = K - 1;
    const int pm = (m == ell) ? sp : m;
    const DModulus M = mods[pm];
    const u64 *key = MODE == 0 ? items[b].key : shared_key;
    const u64 *tg = MODE == 0 ? nullptr : target + (size_t)b * ell * N;
    const u64 *ex = ext + (size_t)b * ell * ell * N;
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    Acc128 a0[2], a1[2];
    u64 r0[2] = { 0, 0 }, r1[2] = { 0, 0 };
#pragma unroll
    for (int e = 0; e < 2; e++) a0[e].clear(), a1[e].clear();
    for (int j = 0; j < ell; j++) {
        u64x2 x;
        if (j == m && MODE == 0) { // an aligned pair of outputs reads an aligned pair of inputs, possibly swapped
            const u32 g = galois_src((u32)k, items[b].elt, logN);
            const u64x2 v = *reinterpret_cast<const u64x2 *>(items[b].src.limb(1, j, N) + (g & ~1u));
            x = (g & 1u) ? u64x2{ v.y, v.x } : v;
        } else {
            const u64 *op = (j == m) ? tg + (size_t)j * N : ex + ((size_t)j * ell + (m < j ? m : m - 1)) * N;
            x = *reinterpret_cast<const u64x2 *>(op + k);
        }
        const u64x2 y0 = *reinterpret_cast<const u64x2 *>(key + (((size_t)j * 2 + 0) * K + pm) * N + k);
        const u64x2 y1 = *reinterpret_cast<const u64x2 *>(key + (((size_t)j * 2 + 1) * K + pm) * N + k);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            a0[e].mac(x[e], y0[e]);
            a1[e].mac(x[e], y1[e]);
        }
        if ((j & 15) == 15) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                r0[e] = addmod(r0[e], a0[e].reduce(M), M.q);
                r1[e] = addmod(r1[e], a1[e].reduce(M), M.q);
                a0[e].clear(), a1[e].clear();
            }
        }
    }
    u64x2 o0, o1;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        o0[e] = addmod(r0[e], a0[e].reduce(M), M.q);
        o1[e] = addmod(r1[e], a1[e].reduce(M), M.q);
    }
    u64 *ac = acc + (size_t)b * 2 * (ell + 1) * N;
    *reinterpret_cast<u64x2 *>(ac + ((size_t)0 * (ell + 1) + m) * N + k) = o0;
    *reinterpret_cast<u64x2 *>(ac + ((size_t)1 * (ell + 1) + m) * N + k) = o1;
}

static bool fuse_mac() { return option(OPT_KS_FUSE_MAC) != 0; }
// work (in 1024-coefficient tiles of lifted digits) from which a key switch takes the large-batch launch sequence
static long big_threshold() { return (long)option(OPT_KS_BIG_TILES); }
// ... and up to which the second NTT phase, the inner products and the special prime's first inverse phase stay one launch
// (always, by default: at N = 2^16 / 24 primes the fused launch is 17 % faster than three, at the HEVM sizes it is even)
static long fuse_mac_threshold() { return (long)option(OPT_KS_FUSE_MAC_TILES); }

// L2..L7 of the key-switch pipeline (fused_ks.hip) once the digits' inverse ROWS phase (L1) has been issued
// (with option ks_fuse_mac_tiles set, a linked step could fall on the un-fused middle, whose last kernel has no continuation form)
bool chain_fusion_supported() { return fuse_mac() && fuse_mac_threshold() == (1L << 40); }

// `digits`: output of the inverse ROWS phase of the key-switch target [B][l][N] (w.digits, or the buffer a fused producer filled)
template <int MODE>
static void b_ks_tail(Context &c, const BatchWs &w, u64 *digits, const KsItem *items, const void *final_items, const u64 *shared_key, int B,
                      int ell, hipStream_t s, const Handoff &h)
{
    const size_t N = c.N;
    const int K = c.K, sp = K - 1;
    // small batches: one launch that recomputes the inverse COLS phase per target modulus (latency); large batches: run it
    // once per limb, then a base-change + forward launch (25-35 % less work in these two steps)
    const long tiles = (long)(N >> 10) * B * ell * ell;
    const bool big = tiles >= big_threshold(), fused_mac = fuse_mac() && tiles < fuse_mac_threshold();
    if (big) {
        launch_ntt_cols_inv(c, digits, (long)N, B * ell, nullptr, 0, ell, s);
        f_ks_lift_fcols(c, digits, w.ext, B, ell, s);
    } else
        f_ks_icols_lift_fcols(c, digits, w.ext, B, ell, s);
    u64 *acc_last = w.acc + (size_t)ell * N;
    const long acc_ps = (long)(ell + 1) * (long)N;
    if (!fused_mac) {
        launch_ntt_rows_fwd(c, w.ext, (long)N, B * ell * ell, c.ks_prime_idx(ell), 0, ell * ell, s);
        DC_LAUNCH(b_ks_mac_kernel<MODE>, dim3((unsigned)(N / (2 * kBT)), ell + 1, B), dim3(kBT), 0, s, w.acc, w.ext, w.target, items,
                           shared_key, ell, K, N, c.logN, c.d_mods);
        f_irows_strided(c, acc_last, acc_ps, sp, 1, acc_last, acc_ps, 2 * B, s);
    } else // MODE 1: the operand a1*b1 is recomputed from the MulItem table (no tensor launch, see b_mul_relin)
        f_ks_frows_mac(c, MODE, w.ext, MODE == 1 ? nullptr : w.target, MODE == 1 ? reinterpret_cast<const KsItem *>(final_items) : items,
                       shared_key, w.acc, B, ell, s, MODE == 0);
    if (big) {
        launch_ntt_cols_inv(c, acc_last, acc_ps, 2 * B, nullptr, sp, 1, s);
        f_dr_lift_fcols(c, acc_last, acc_ps, w.tmp, 2 * B, ell, sp, s);
    } else
        f_dr_icols_lift_fcols(c, acc_last, acc_ps, w.tmp, 2 * B, ell, sp, s);
    f_frows_final(c, (MODE == 1 && fused_mac) ? 4 : MODE, w.tmp, final_items, w.acc, 2 * B, ell, sp, s, RsItem{}, nullptr, nullptr, h,
                  MODE == 0 && fused_mac); // (a rotation's base term was folded into the fused middle's accumulators)
}

void b_rotate_hops(Context &c, const BatchWs &w, const KsItem *d_items, int B, int ell, hipStream_t s, const Handoff &h, int unique)
{
    if (c.hybrid()) { // (the plan links no steps in this mode: h is empty)
        hyb_rotate_hops(c, w, d_items, B, ell, s, unique);
        return;
    }
    f_irows_rot_c1(c, d_items, ell, w.digits, B, s);
    b_ks_tail<0>(c, w, w.digits, d_items, d_items, nullptr, B, ell, s, h);
}

void b_mul_relin(Context &c, const BatchWs &w, const MulItem *d_items, const u64 *relin_key, int B, int ell, hipStream_t s, const Handoff &h)
{
    if (c.hybrid()) {
        hyb_mul_relin(c, w, d_items, relin_key, B, ell, s);
        return;
    }
    const size_t N = c.N;
    const bool fused_mac = fuse_mac() && (long)(N >> 10) * B * ell * ell < fuse_mac_threshold(); // same split as b_ks_tail
    u64 *digits = w.digits;
    if (h.in) // a fused producer already ran the inverse ROWS phase of a1*b1 (the plan only links steps when fuse_mac() holds)
        digits = const_cast<u64 *>(h.in);
    else if (!fused_mac) {
        DC_LAUNCH(b_tensor_kernel, dim3((unsigned)(N / (2 * kBT)), ell, B), dim3(kBT), 0, s, d_items, w.target, ell, N, c.d_mods);
        f_irows_strided(c, w.target, (long)N, 0, ell, w.digits, (long)N, B * ell, s);
    } else // small batches: a1*b1 is formed in the loaders and a0*b0, a0*b1 + a1*b0 in the last kernel's epilogue
        f_irows_tensor_c2(c, d_items, ell, w.digits, B, s);
    b_ks_tail<1>(c, w, digits, nullptr, d_items, relin_key, B, ell, s, h);
}

void b_rescale(Context &c, const BatchWs &w, const RsItem *d_items, int B, int ell, hipStream_t s, const SumSrc *d_srcs, const Handoff &h)
{
    const int l = ell - 1;
    const u64 *last = w.digits; // [B][2][N]
    if (h.in)
        last = h.in;
    else
        f_irows_rs_last(c, d_items, d_srcs, l, w.digits, B, s);
    f_dr_icols_lift_fcols(c, last, (long)c.N, w.tmp, 2 * B, l, l, s);
    f_frows_final(c, 2, w.tmp, d_items, nullptr, 2 * B, l, l, s, RsItem{}, nullptr, d_srcs, h);
}

void rescale_fused(Context &c, const Workspace &w, CtView dst, CtView src, int ell, const u64 *plain, hipStream_t s)
{
    const int l = ell - 1;
    u64 *last = w.ks_digits; // [2][N]
    f_irows_rs_single(c, src, l, last, s);
    f_dr_icols_lift_fcols(c, last, (long)c.N, w.ks_tmp, 2, l, l, s);
    f_frows_final(c, 3, w.ks_tmp, nullptr, nullptr, 2, l, l, s, RsItem{ src, dst, 0, 0, nullptr, nullptr }, plain);
}

// ---- limb-wise batched kernels ---------------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(kBT) void b_ew_kernel(const EwItem *__restrict__ items, int polys, int b_polys, size_t N,
                                                    const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, p = blockIdx.z % polys, b = blockIdx.z / polys;
    const EwItem it = items[b];
    const DModulus m = mods[i];
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    const u64x2 va = *reinterpret_cast<const u64x2 *>(it.a.limb(p, i, N) + k);
    u64x2 r = va;
    if (OP == (int)EwOp::Neg) {
        r.x = negmod(va.x, m.q), r.y = negmod(va.y, m.q);
    } else if (OP == (int)EwOp::Mul) {
        const u64x2 vb = *reinterpret_cast<const u64x2 *>(it.b.limb(b_polys == 1 ? 0 : p, i, N) + k);
        r.x = mulmod(va.x, vb.x, m), r.y = mulmod(va.y, vb.y, m);
    }
    *reinterpret_cast<u64x2 *>(it.dst.limb(p, i, N) + k) = r;
}

void b_ew(Context &c, EwOp op, const EwItem *d_items, int B, int polys, int b_polys, int ell, hipStream_t s)
{
    dim3 grid((unsigned)(c.N / (2 * kBT)), (unsigned)ell, (unsigned)(polys * B)), block(kBT);
    switch (op) {
    case EwOp::Neg: DC_LAUNCH(b_ew_kernel<2>, grid, block, 0, s, d_items, polys, b_polys, c.N, c.d_mods); break;
    case EwOp::Mul: DC_LAUNCH(b_ew_kernel<3>, grid, block, 0, s, d_items, polys, b_polys, c.N, c.d_mods); break;
    case EwOp::Copy: DC_LAUNCH(b_ew_kernel<4>, grid, block, 0, s, d_items, polys, b_polys, c.N, c.d_mods); break;
    default: fprintf(stderr, "[dacapo_amd] b_ew: unsupported op\n"); abort();
    }
}

// dst.c0 = a.c0 + plain, dst.c1 = a.c1.  grid = (N/512, l, 2B)
__global__ __launch_bounds__(kBT) void b_add_plain_kernel(const EwItem *__restrict__ items, size_t N, const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, p = blockIdx.z & 1, b = blockIdx.z >> 1;
    const EwItem it = items[b];
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    u64x2 v = *reinterpret_cast<const u64x2 *>(it.a.limb(p, i, N) + k);
    if (p == 0) {
        const u64 q = mods[i].q;
        const u64x2 wv = *reinterpret_cast<const u64x2 *>(it.b.p + (size_t)i * N + k);
        v.x = addmod(v.x, wv.x, q);
        v.y = addmod(v.y, wv.y, q);
    }
    *reinterpret_cast<u64x2 *>(it.dst.limb(p, i, N) + k) = v;
}

void b_add_plain(Context &c, const EwItem *d_items, int B, int ell, hipStream_t s)
{
    DC_LAUNCH(b_add_plain_kernel, dim3((unsigned)(c.N / (2 * kBT)), (unsigned)ell, (unsigned)(2 * B)), dim3(kBT), 0, s, d_items,
                       c.N, c.d_mods);
}

// dst = sum of the item's terms; a term is a ciphertext or a ciphertext times a plaintext (a single-use mulcp result
// folded in).  Modular addition and multiplication are exact, so the limbs equal those of the reference's sequence
// multiply_plain ... add ... add in any order.  Products accumulate in 128 bits (reduced every 16), plain terms lazily.
// grid = (N/512, l, 2B)
__global__ __launch_bounds__(kBT) void b_sum_kernel(const SumItem *__restrict__ items, const SumSrc *__restrict__ srcs, size_t N,
                                                     const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, p = blockIdx.z & 1, b = blockIdx.z >> 1;
    const SumItem it = items[b];
    const DModulus M = mods[i];
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    u64 s0 = 0, s1 = 0; // lazy: every term < 2^60, folded every 8 terms
    Acc128 a0, a1;
    a0.clear(), a1.clear();
    int n_plain = 0, n_prod = 0;
    for (int t = 0; t < it.count; t++) {
        const SumSrc src = srcs[it.first + t];
        const u64x2 v = *reinterpret_cast<const u64x2 *>(src.v.limb(p, i, N) + k);
        if (src.plain) {
            const u64x2 w = *reinterpret_cast<const u64x2 *>(src.plain + (size_t)i * N + k);
            a0.mac(v.x, w.x);
            a1.mac(v.y, w.y);
            if ((++n_prod & 15) == 0) {
                s0 = fold60(s0, M.delta) + a0.reduce(M);
                s1 = fold60(s1, M.delta) + a1.reduce(M);
                a0.clear(), a1.clear();
            }
        } else {
            s0 += v.x;
            s1 += v.y;
            if ((++n_plain & 7) == 0) {
                s0 = fold60(s0, M.delta);
                s1 = fold60(s1, M.delta);
            }
        }
    }
    u64x2 r;
    r.x = addmod(canon(s0, M), a0.reduce(M), M.q);
    r.y = addmod(canon(s1, M), a1.reduce(M), M.q);
    *reinterpret_cast<u64x2 *>(it.dst.limb(p, i, N) + k) = r;
}

// ---- ModRaise -----------------------------------------------------------------------------------------------------------------
// scratch[b*2 + p][k] = limb 0 of polynomial p of item b (NTT form; the caller then takes it to the coefficient domain).
// items == nullptr: one item passed by value (the one-instruction-at-a-time loop: no allocation, no copy, no synchronisation)
__global__ __launch_bounds__(kBT) void b_modraise_gather_kernel(u64 *__restrict__ scratch, const EwItem *__restrict__ items, EwItem single, size_t N)
{
    const int z = blockIdx.y, b = z >> 1, p = z & 1;
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    const CtView a = items ? items[b].a : single.a;
    *reinterpret_cast<u64x2 *>(scratch + (size_t)z * N + k) = *reinterpret_cast<const u64x2 *>(a.limb(p, 0, N) + k);
}
// dst[p][i][k] = centred(scratch[b*2+p][k] mod q_0) mod q_i : the integer in (-q_0/2, q_0/2] read modulo every prime of the target level
// grid = (N/512, target, 2B)
__global__ __launch_bounds__(kBT) void b_modraise_lift_kernel(const u64 *__restrict__ scratch, const EwItem *__restrict__ items, EwItem single,
                                                               size_t N, const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, z = blockIdx.z, b = z >> 1, p = z & 1;
    const CtView dstv = items ? items[b].dst : single.dst;
    const DModulus Mi = mods[i];
    const u64 q0 = mods[0].q, qi = Mi.q, half = q0 >> 1;
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    const u64x2 v = *reinterpret_cast<const u64x2 *>(scratch + (size_t)z * N + k);
    u64x2 r;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        if (v[e] > half) { // negative: -(q0 - v) mod q_i (one conditional subtraction within a width class: recanon)
            const u64 m = recanon(q0 - v[e], Mi);
            r[e] = m ? qi - m : 0;
        } else
            r[e] = recanon(v[e], Mi);
    }
    *reinterpret_cast<u64x2 *>(dstv.limb(p, i, N) + k) = r;
}

void modraise(Context &c, u64 *scratch, const EwItem *h_items, int B, int target, hipStream_t s, const EwItem *d_items)
{
    const size_t N = c.N;
    if (!d_items && B != 1) {
        fprintf(stderr, "[dacapo_amd] modraise: a batch needs its item table on the device\n");
        abort();
    }
    const EwItem single = d_items ? EwItem{} : h_items[0]; // eager path: the one item travels as a kernel argument
    DC_LAUNCH(b_modraise_gather_kernel, dim3((unsigned)(N / (2 * kBT)), (unsigned)(2 * B)), dim3(kBT), 0, s, scratch, d_items, single, N);
    launch_ntt(c, true, scratch, (long)N, 2 * B, nullptr, 0, 1, s);
    DC_LAUNCH(b_modraise_lift_kernel, dim3((unsigned)(N / (2 * kBT)), (unsigned)target, (unsigned)(2 * B)), dim3(kBT), 0, s, scratch,
                       d_items, single, N, c.d_mods);
    for (int b = 0; b < B; b++) { // one forward transform over both polynomials of an item: [2][target] limbs, the register's poly stride apart
        const CtView d = h_items[b].dst;
        if (d.poly_stride == (long)target * (long)N)
            launch_ntt(c, false, d.limb(0, 0, N), (long)N, 2 * target, nullptr, 0, target, s);
        else
            for (int p = 0; p < 2; p++) launch_ntt(c, false, d.limb(p, 0, N), (long)N, target, nullptr, 0, 0, s);
    }
}

// The same sum with both polynomials of an item in one thread: a plaintext limb is read once instead of twice (3 instead of 4 loads per
// product term).  grid = (N/512, l, B); taken for launches large enough to fill the chip with half the workgroups.
__global__ __launch_bounds__(kBT) void b_sum_pair_kernel(const SumItem *__restrict__ items, const SumSrc *__restrict__ srcs, size_t N,
                                                          const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, b = blockIdx.z;
    const SumItem it = items[b];
    const DModulus M = mods[i];
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    u64 s[2][2] = { { 0, 0 }, { 0, 0 } }; // [poly][element], lazy exactly as in b_sum_kernel
    Acc128 a[2][2];
#pragma unroll
    for (int p = 0; p < 2; p++) a[p][0].clear(), a[p][1].clear();
    int n_plain = 0, n_prod = 0;
    // (term t + 1's three loads are issued before term t's products: every term costs a dependent chain item table -> pointers -> limbs, and
    // with one term in flight per thread the kernel waited on it)
    // (and term t + 2's table entry with them)
    auto fetch = [&](const SumSrc &src, const u64 *&pl, u64x2 &v0, u64x2 &v1, u64x2 &w) {
        pl = src.plain;
        v0 = *reinterpret_cast<const u64x2 *>(src.v.limb(0, i, N) + k);
        v1 = *reinterpret_cast<const u64x2 *>(src.v.limb(1, i, N) + k);
        if (pl) w = *reinterpret_cast<const u64x2 *>(pl + (size_t)i * N + k);
    };
    const u64 *pln = nullptr;
    u64x2 v0n{}, v1n{}, wn{};
    SumSrc sn{};
    if (it.count > 0) fetch(srcs[it.first], pln, v0n, v1n, wn);
    if (it.count > 1) sn = srcs[it.first + 1];
    for (int t = 0; t < it.count; t++) {
        const u64 *plain = pln;
        const u64x2 v0 = v0n, v1 = v1n, w = wn;
        if (t + 1 < it.count) {
            fetch(sn, pln, v0n, v1n, wn);
            if (t + 2 < it.count) sn = srcs[it.first + t + 2];
        }
        if (plain) {
            a[0][0].mac(v0.x, w.x), a[0][1].mac(v0.y, w.y);
            a[1][0].mac(v1.x, w.x), a[1][1].mac(v1.y, w.y);
            if ((++n_prod & 15) == 0) {
#pragma unroll
                for (int p = 0; p < 2; p++)
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        s[p][e] = fold60(s[p][e], M.delta) + a[p][e].reduce(M);
                        a[p][e].clear();
                    }
            }
        } else {
            s[0][0] += v0.x, s[0][1] += v0.y, s[1][0] += v1.x, s[1][1] += v1.y;
            if ((++n_plain & 7) == 0) {
#pragma unroll
                for (int p = 0; p < 2; p++) s[p][0] = fold60(s[p][0], M.delta), s[p][1] = fold60(s[p][1], M.delta);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
        u64x2 r;
        r.x = addmod(canon(s[p][0], M), a[p][0].reduce(M), M.q);
        r.y = addmod(canon(s[p][1], M), a[p][1].reduce(M), M.q);
        *reinterpret_cast<u64x2 *>(it.dst.limb(p, i, N) + k) = r;
    }
}

// Sums of one step that share sources, up to kSumGroup items to a thread (plan.hpp SumGroup): the group's sources are walked once, every
// ciphertext limb read once for all the items that use it; the plaintexts (one per item and source) are the only per-item reads.  One
// coefficient per thread (8 items x 2 polynomials of 128-bit accumulators are the register budget); grid = (N/256, groups, l).
// Exact arithmetic: every destination receives the canonical residue of the same integer sum as b_sum_kernel's.
__global__ __launch_bounds__(kBT) void b_sum_group_kernel(const SumGroup *__restrict__ groups, const SumGroupSrc *__restrict__ gsrcs, size_t N,
                                                           const DModulus *__restrict__ mods)
{
    // (groups fastest: the groups of a step mostly share their sources too -- two groups of 7 giant steps over the same 16 baby steps)
    const int i = blockIdx.z;
    const SumGroup &g = groups[blockIdx.y];
    const DModulus M = mods[i];
    const size_t k = (size_t)blockIdx.x * kBT + threadIdx.x;
    const int first = g.first, count = g.count, items = g.items;
    Acc128 a[kSumGroup][2];
#pragma unroll
    for (int b = 0; b < kSumGroup; b++) a[b][0].clear(), a[b][1].clear();
    // (source t + 1's loads are issued before source t's products, as in b_sum_pair_kernel)
    auto fetch = [&](int t, u64 &v0, u64 &v1, u64 (&w)[kSumGroup], unsigned &mul, unsigned &add) {
        const SumGroupSrc &src = gsrcs[first + t];
        v0 = src.v.limb(0, i, N)[k], v1 = src.v.limb(1, i, N)[k];
        mul = add = 0;
#pragma unroll
        for (int b = 0; b < kSumGroup; b++) {
            const u64 *p = src.plain[b]; // wave-uniform
            if (p == sum_add_only())
                add |= 1u << b;
            else if (p)
                mul |= 1u << b, w[b] = p[(size_t)i * N + k];
        }
    };
    u64 v0n = 0, v1n = 0, wn[kSumGroup];
    unsigned muln = 0, addn = 0;
#pragma unroll
    for (int b = 0; b < kSumGroup; b++) wn[b] = 0;
    if (count > 0) fetch(0, v0n, v1n, wn, muln, addn);
    for (int t = 0; t < count; t++) {
        const u64 v0 = v0n, v1 = v1n;
        const unsigned mul = muln, add = addn;
        u64 w[kSumGroup];
#pragma unroll
        for (int b = 0; b < kSumGroup; b++) w[b] = wn[b];
        if (t + 1 < count) fetch(t + 1, v0n, v1n, wn, muln, addn);
#pragma unroll
        for (int b = 0; b < kSumGroup; b++) {
            if (mul >> b & 1u)
                a[b][0].mac(v0, w[b]), a[b][1].mac(v1, w[b]);
            else if (add >> b & 1u)
                a[b][0].lo += v0, a[b][0].hi += a[b][0].lo < v0, a[b][1].lo += v1, a[b][1].hi += a[b][1].lo < v1;
        }
        if ((t & 15) == 15 && t + 1 < count) { // a 128-bit accumulator holds 16 products of canonical residues: fold it into a word
#pragma unroll
            for (int b = 0; b < kSumGroup; b++)
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const u64 f = a[b][p].reduce(M);
                    a[b][p].clear(), a[b][p].lo = f; // (counts as one more, tiny, term: 16 products + 2^60 < 2^124)
                }
        }
    }
#pragma unroll
    for (int b = 0; b < kSumGroup; b++)
        if (b < items) {
            g.dst[b].limb(0, i, N)[k] = a[b][0].reduce(M);
            g.dst[b].limb(1, i, N)[k] = a[b][1].reduce(M);
        }
}

void b_sum_group(Context &c, const SumGroup *d_groups, const SumGroupSrc *d_gsrcs, int G, int ell, hipStream_t s)
{
    DC_LAUNCH(b_sum_group_kernel, dim3((unsigned)(c.N / kBT), (unsigned)G, (unsigned)ell), dim3(kBT), 0, s, d_groups, d_gsrcs, c.N, c.d_mods);
}

static long sum_pair_min_workgroups()
{ // option sum_pair_min_wgs: workgroups the paired form must still have (default 4 per CU); 0 = always, a huge value = never
    return (long)option(OPT_SUM_PAIR_MIN_WGS);
}

void b_sum(Context &c, const SumItem *d_items, const SumSrc *d_srcs, int B, int ell, hipStream_t s)
{
    if ((long)(c.N / (2 * kBT)) * ell * B >= sum_pair_min_workgroups()) {
        DC_LAUNCH(b_sum_pair_kernel, dim3((unsigned)(c.N / (2 * kBT)), (unsigned)ell, (unsigned)B), dim3(kBT), 0, s, d_items, d_srcs, c.N,
                           c.d_mods);
        return;
    }
    DC_LAUNCH(b_sum_kernel, dim3((unsigned)(c.N / (2 * kBT)), (unsigned)ell, (unsigned)(2 * B)), dim3(kBT), 0, s, d_items, d_srcs,
                       c.N, c.d_mods);
}

} // namespace dacapo
