// Batched forms of the composite evaluator ops (plan.hpp): B independent ciphertext ops of one kind and level per
// launch sequence.  Same arithmetic, same order of modular operations per limb as ckks_ops.hip -- only the grid grows
// by the batch dimension and operands come from a device table of views.  Algorithms: SEAL 4.0
// Evaluator::switch_key_inplace / rescale_to_next / multiply [SEAL-upstream], reached from SEAL_HEVM.cpp:273,283,315-316.
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

// rotation prologue: c0perm[b] = perm(src.c0), target[b] = digits[b] = perm(src.c1).  grid = (N/256, l, 2B)
__global__ __launch_bounds__(kBT) void b_galois_kernel(const KsItem *__restrict__ items, u64 *__restrict__ c0perm,
                                                        u64 *__restrict__ target, u64 *__restrict__ digits, int ell, int logN)
{
    const size_t N = (size_t)1 << logN;
    const int i = blockIdx.y, p = blockIdx.z & 1, b = blockIdx.z >> 1;
    const KsItem it = items[b];
    const u32 k = blockIdx.x * kBT + threadIdx.x;
    const u64 v = it.src.limb(p, i, N)[galois_src(k, it.elt, logN)];
    const size_t o = ((size_t)b * ell + i) * N + k;
    if (p == 0)
        c0perm[o] = v;
    else {
        target[o] = v;
        digits[o] = v;
    }
}

// ckks_multiply prologue: dst.c0 = a0 b0, dst.c1 = a0 b1 + a1 b0, target[b] = digits[b] = a1 b1.  grid = (N/512, l, B)
__global__ __launch_bounds__(kBT) void b_tensor_kernel(const MulItem *__restrict__ items, u64 *__restrict__ target,
                                                        u64 *__restrict__ digits, int ell, size_t N,
                                                        const DModulus *__restrict__ mods)
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
    *reinterpret_cast<u64x2 *>(digits + o) = c2;
}

// digit j of item b reduced into its e-th other modulus.  grid = (N/512, l [e], B*l [b,j])
__global__ __launch_bounds__(kBT) void b_ks_lift_kernel(u64 *__restrict__ ext, const u64 *__restrict__ digits, int ell, int sp,
                                                         size_t N, const DModulus *__restrict__ mods)
{
    const int e = blockIdx.y, z = blockIdx.z, j = z % ell;
    const u64 qm = mods[ks_other_prime(j, e, ell, sp)].q;
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    u64x2 v = *reinterpret_cast<const u64x2 *>(digits + (size_t)z * N + k);
    v.x = v.x >= qm ? v.x - qm : v.x;
    v.y = v.y >= qm ? v.y - qm : v.y;
    *reinterpret_cast<u64x2 *>(ext + ((size_t)z * ell + e) * N + k) = v;
}

// inner products with the key of item b.  grid = (N/512, l+1 [m], B)
__global__ __launch_bounds__(kBT) void b_ks_mac_kernel(u64 *__restrict__ acc, const u64 *__restrict__ ext,
                                                        const u64 *__restrict__ target, const KsItem *__restrict__ items,
                                                        const u64 *__restrict__ shared_key, int ell, int K, size_t N,
                                                        const DModulus *__restrict__ mods)
{
    const int m = blockIdx.y, b = blockIdx.z, sp = K - 1;
    const int pm = (m == ell) ? sp : m;
    const DModulus M = mods[pm];
    const u64 *key = items ? items[b].key : shared_key;
    const u64 *tg = target + (size_t)b * ell * N;
    const u64 *ex = ext + (size_t)b * ell * ell * N;
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    Acc128 a0[2], a1[2];
    u64 r0[2] = { 0, 0 }, r1[2] = { 0, 0 };
#pragma unroll
    for (int e = 0; e < 2; e++) a0[e].clear(), a1[e].clear();
    for (int j = 0; j < ell; j++) {
        const u64 *op = (j == m) ? tg + (size_t)j * N : ex + ((size_t)j * ell + (m < j ? m : m - 1)) * N;
        const u64x2 x = *reinterpret_cast<const u64x2 *>(op + k);
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

// divide-and-round, first half, over 2B polynomials.  grid = (N/512, cnt, 2B)
__global__ __launch_bounds__(kBT) void b_dr_lift_kernel(u64 *__restrict__ tmp, long tmp_ps, const u64 *__restrict__ last, long last_ps,
                                                         int l, int K, size_t N, const DModulus *__restrict__ mods,
                                                         const u64 *__restrict__ half_mod)
{
    const int i = blockIdx.y, p = blockIdx.z;
    const u64 ql = mods[l].q, qi = mods[i].q, half = ql >> 1;
    const u64 neg_half = qi - half_mod[(size_t)l * K + i];
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    const u64x2 v = *reinterpret_cast<const u64x2 *>(last + p * last_ps + k);
    u64x2 r;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        u64 y = v[e] + half;
        y = y >= ql ? y - ql : y;
        y = y >= qi ? y - qi : y;
        y += neg_half;
        r[e] = y >= qi ? y - qi : y;
    }
    *reinterpret_cast<u64x2 *>(tmp + p * tmp_ps + (size_t)i * N + k) = r;
}

// key-switch epilogue: out[p][i] = base[p][i] + (acc[p][i] - tmp[p][i]) P^{-1}.  MODE 0: rotation (base0 = c0perm, base1 = 0),
// MODE 1: relinearisation (base = the tensor product already sitting in dst).  grid = (N/512, l, 2B)
template <int MODE>
__global__ __launch_bounds__(kBT) void b_ks_final_kernel(const void *__restrict__ items_, const u64 *__restrict__ acc,
                                                          const u64 *__restrict__ tmp, const u64 *__restrict__ c0perm, int ell,
                                                          int K, size_t N, const DModulus *__restrict__ mods,
                                                          const u64 *__restrict__ inv_last)
{
    const int i = blockIdx.y, p = blockIdx.z & 1, b = blockIdx.z >> 1, sp = K - 1;
    const DModulus M = mods[i];
    const u64 inv = inv_last[(size_t)sp * K + i];
    CtView dst;
    if (MODE == 0)
        dst = reinterpret_cast<const KsItem *>(items_)[b].dst;
    else
        dst = reinterpret_cast<const MulItem *>(items_)[b].dst;
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    const u64x2 xv = *reinterpret_cast<const u64x2 *>(acc + (((size_t)b * 2 + p) * (ell + 1) + i) * N + k);
    const u64x2 tv = *reinterpret_cast<const u64x2 *>(tmp + (((size_t)b * 2 + p) * ell + i) * N + k);
    u64x2 bv = { 0, 0 };
    if (MODE == 1)
        bv = *reinterpret_cast<const u64x2 *>(dst.limb(p, i, N) + k);
    else if (p == 0)
        bv = *reinterpret_cast<const u64x2 *>(c0perm + ((size_t)b * ell + i) * N + k);
    u64x2 r;
#pragma unroll
    for (int e = 0; e < 2; e++) r[e] = addmod(bv[e], mulmod(submod(xv[e], tv[e], M.q), inv, M), M.q);
    *reinterpret_cast<u64x2 *>(dst.limb(p, i, N) + k) = r;
}

// steps shared by rotation and relinearisation once target/digits are in place
static void b_ks_core(Context &c, const BatchWs &w, const KsItem *items, const u64 *shared_key, int B, int ell, hipStream_t s)
{
    const size_t N = c.N;
    const int K = c.K, sp = K - 1;
    const unsigned gx = (unsigned)(N / (2 * kBT));
    launch_ntt(c, true, w.digits, (long)N, B * ell, nullptr, 0, ell, s);
    hipLaunchKernelGGL(b_ks_lift_kernel, dim3(gx, ell, B * ell), dim3(kBT), 0, s, w.ext, w.digits, ell, sp, N, c.d_mods);
    launch_ntt(c, false, w.ext, (long)N, B * ell * ell, c.ks_prime_idx(ell), 0, ell * ell, s);
    hipLaunchKernelGGL(b_ks_mac_kernel, dim3(gx, ell + 1, B), dim3(kBT), 0, s, w.acc, w.ext, w.target, items, shared_key, ell, K, N,
                       c.d_mods);
    u64 *acc_last = w.acc + (size_t)ell * N;
    const long acc_ps = (long)(ell + 1) * (long)N;
    launch_ntt(c, true, acc_last, acc_ps, 2 * B, nullptr, sp, 1, s);
    hipLaunchKernelGGL(b_dr_lift_kernel, dim3(gx, ell, 2 * B), dim3(kBT), 0, s, w.tmp, (long)ell * (long)N, acc_last, acc_ps, sp, K, N,
                       c.d_mods, c.d_half_mod);
    launch_ntt(c, false, w.tmp, (long)N, 2 * B * ell, nullptr, 0, ell, s);
}

void b_rotate_hops(Context &c, const BatchWs &w, const KsItem *d_items, int B, int ell, hipStream_t s)
{
    const size_t N = c.N;
    hipLaunchKernelGGL(b_galois_kernel, dim3((unsigned)(N / kBT), ell, 2 * B), dim3(kBT), 0, s, d_items, w.c0perm, w.target, w.digits,
                       ell, c.logN);
    b_ks_core(c, w, d_items, nullptr, B, ell, s);
    hipLaunchKernelGGL(b_ks_final_kernel<0>, dim3((unsigned)(N / (2 * kBT)), ell, 2 * B), dim3(kBT), 0, s, (const void *)d_items,
                       w.acc, w.tmp, w.c0perm, ell, c.K, N, c.d_mods, c.d_inv_last);
}

void b_mul_relin(Context &c, const BatchWs &w, const MulItem *d_items, const u64 *relin_key, int B, int ell, hipStream_t s)
{
    const size_t N = c.N;
    hipLaunchKernelGGL(b_tensor_kernel, dim3((unsigned)(N / (2 * kBT)), ell, B), dim3(kBT), 0, s, d_items, w.target, w.digits, ell, N,
                       c.d_mods);
    b_ks_core(c, w, nullptr, relin_key, B, ell, s);
    hipLaunchKernelGGL(b_ks_final_kernel<1>, dim3((unsigned)(N / (2 * kBT)), ell, 2 * B), dim3(kBT), 0, s, (const void *)d_items,
                       w.acc, w.tmp, (const u64 *)nullptr, ell, c.K, N, c.d_mods, c.d_inv_last);
}

// rescale: last[b][p] = src limb l of poly p.  grid = (N/512, 1, 2B)
__global__ __launch_bounds__(kBT) void b_rs_copy_kernel(const RsItem *__restrict__ items, u64 *__restrict__ last, int l, size_t N)
{
    const int p = blockIdx.z & 1, b = blockIdx.z >> 1;
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    *reinterpret_cast<u64x2 *>(last + (size_t)blockIdx.z * N + k) = *reinterpret_cast<const u64x2 *>(items[b].src.limb(p, l, N) + k);
}
// dst[p][i] = (src[p][i] - tmp[p][i]) q_l^{-1}.  grid = (N/512, l, 2B)
__global__ __launch_bounds__(kBT) void b_rs_final_kernel(const RsItem *__restrict__ items, const u64 *__restrict__ tmp, int l, int K,
                                                          size_t N, const DModulus *__restrict__ mods,
                                                          const u64 *__restrict__ inv_last)
{
    const int i = blockIdx.y, p = blockIdx.z & 1, b = blockIdx.z >> 1;
    const DModulus M = mods[i];
    const u64 inv = inv_last[(size_t)l * K + i];
    const RsItem it = items[b];
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    const u64x2 xv = *reinterpret_cast<const u64x2 *>(it.src.limb(p, i, N) + k);
    const u64x2 tv = *reinterpret_cast<const u64x2 *>(tmp + ((size_t)blockIdx.z * l + i) * N + k);
    u64x2 r;
#pragma unroll
    for (int e = 0; e < 2; e++) r[e] = mulmod(submod(xv[e], tv[e], M.q), inv, M);
    *reinterpret_cast<u64x2 *>(it.dst.limb(p, i, N) + k) = r;
}

void b_rescale(Context &c, const BatchWs &w, const RsItem *d_items, int B, int ell, hipStream_t s)
{
    const size_t N = c.N;
    const int l = ell - 1;
    const unsigned gx = (unsigned)(N / (2 * kBT));
    u64 *last = w.digits; // [B][2][N]
    hipLaunchKernelGGL(b_rs_copy_kernel, dim3(gx, 1, 2 * B), dim3(kBT), 0, s, d_items, last, l, N);
    launch_ntt(c, true, last, (long)N, 2 * B, nullptr, l, 1, s);
    hipLaunchKernelGGL(b_dr_lift_kernel, dim3(gx, l, 2 * B), dim3(kBT), 0, s, w.tmp, (long)l * (long)N, last, (long)N, l, c.K, N,
                       c.d_mods, c.d_half_mod);
    launch_ntt(c, false, w.tmp, (long)N, 2 * B * l, nullptr, 0, l, s);
    hipLaunchKernelGGL(b_rs_final_kernel, dim3(gx, l, 2 * B), dim3(kBT), 0, s, d_items, w.tmp, l, c.K, N, c.d_mods, c.d_inv_last);
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
    case EwOp::Neg: hipLaunchKernelGGL(b_ew_kernel<2>, grid, block, 0, s, d_items, polys, b_polys, c.N, c.d_mods); break;
    case EwOp::Mul: hipLaunchKernelGGL(b_ew_kernel<3>, grid, block, 0, s, d_items, polys, b_polys, c.N, c.d_mods); break;
    case EwOp::Copy: hipLaunchKernelGGL(b_ew_kernel<4>, grid, block, 0, s, d_items, polys, b_polys, c.N, c.d_mods); break;
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
    hipLaunchKernelGGL(b_add_plain_kernel, dim3((unsigned)(c.N / (2 * kBT)), (unsigned)ell, (unsigned)(2 * B)), dim3(kBT), 0, s, d_items,
                       c.N, c.d_mods);
}

// dst = sum of the item's sources: exact modular additions, so the limbs equal those of any sequential order of the
// reference's ct+ct chain.  grid = (N/512, l, 2B)
__global__ __launch_bounds__(kBT) void b_sum_kernel(const SumItem *__restrict__ items, const CtView *__restrict__ srcs, size_t N,
                                                     const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, p = blockIdx.z & 1, b = blockIdx.z >> 1;
    const SumItem it = items[b];
    const DModulus M = mods[i];
    const size_t k = ((size_t)blockIdx.x * kBT + threadIdx.x) * 2;
    u64 s0 = 0, s1 = 0; // lazy: every term < 2^60, folded every 8 terms
    for (int t = 0; t < it.count; t++) {
        const u64x2 v = *reinterpret_cast<const u64x2 *>(srcs[it.first + t].limb(p, i, N) + k);
        s0 += v.x;
        s1 += v.y;
        if ((t & 7) == 7) {
            s0 = fold60(s0, M.delta);
            s1 = fold60(s1, M.delta);
        }
    }
    u64x2 r;
    r.x = canon(s0, M);
    r.y = canon(s1, M);
    *reinterpret_cast<u64x2 *>(it.dst.limb(p, i, N) + k) = r;
}

void b_sum(Context &c, const SumItem *d_items, const CtView *d_srcs, int B, int ell, hipStream_t s)
{
    hipLaunchKernelGGL(b_sum_kernel, dim3((unsigned)(c.N / (2 * kBT)), (unsigned)ell, (unsigned)(2 * B)), dim3(kBT), 0, s, d_items, d_srcs,
                       c.N, c.d_mods);
}

} // namespace dacapo
