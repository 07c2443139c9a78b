// Composite evaluator operations of the HEVM path, as sequences of stream-ordered launches:
//   keyswitch  = Evaluator::switch_key_inplace          (SEAL_HEVM.cpp:273 via rotate_vector, :316 via relinearize)
//   rescale    = Evaluator::rescale_to_next             (SEAL_HEVM.cpp:283)
//   mul_relin  = Evaluator::multiply + relinearize_inplace (SEAL_HEVM.cpp:315-316)
//   rotate_hop = Evaluator::apply_galois_inplace         (one hop of SEAL_HEVM.cpp:273)
// Algorithms follow SEAL 4.0 [SEAL-upstream evaluator.cpp, rns.cpp]; all intermediate results that SEAL
// defines on canonical residues are canonical here too, so outputs are bit-identical by construction.
#include "kernels.hpp"
#include "plan.hpp"

namespace dacapo {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
constexpr int kOpThreads = 256;

// step 2 of switch_key_inplace: digit j (coefficient domain, canonical mod q_j) reduced into every other
// modulus.  All chain primes lie in (2^60 - 2^28, 2^60), so one conditional subtraction is the full reduction
// (SEAL: modulo_poly_coeffs when q_j > q_m, plain copy otherwise).
__global__ __launch_bounds__(kOpThreads) void ks_lift_kernel(u64 *__restrict__ ext, const u64 *__restrict__ digits, int ell,
                                                              int sp, size_t N, const DModulus *__restrict__ mods)
{
    const int e = blockIdx.y, j = blockIdx.z;
    const DModulus Mm = mods[ks_other_prime(j, e, ell, sp)];
    const size_t k = ((size_t)blockIdx.x * kOpThreads + threadIdx.x) * 2;
    u64x2 v = *reinterpret_cast<const u64x2 *>(digits + (size_t)j * N + k);
    v.x = recanon(v.x, Mm);
    v.y = recanon(v.y, Mm);
    *reinterpret_cast<u64x2 *>(ext + ((size_t)j * ell + e) * N + k) = v;
}

// inner products: acc[kc][m] = sum_j operand(j,m) * key[j][kc][prime(m)], operand = original NTT-form limb when
// m == j, lifted+NTT'd digit otherwise.  grid = (N/512, ell+1).
__global__ __launch_bounds__(kOpThreads) void ks_mac_kernel(u64 *__restrict__ acc, const u64 *__restrict__ ext,
                                                             const u64 *__restrict__ target, const u64 *__restrict__ key,
                                                             int ell, int K, size_t N, const DModulus *__restrict__ mods)
{
    const int m = blockIdx.y, sp = K - 1;
    const int pm = (m == ell) ? sp : m;
    const DModulus M = mods[pm];
    const size_t k = ((size_t)blockIdx.x * kOpThreads + threadIdx.x) * 2;
    Acc128 a0[2], a1[2];
    u64 r0[2] = { 0, 0 }, r1[2] = { 0, 0 };
#pragma unroll
    for (int e = 0; e < 2; e++) a0[e].clear(), a1[e].clear();
    for (int j = 0; j < ell; j++) {
        const u64 *op = (j == m) ? target + (size_t)j * N : ext + ((size_t)j * ell + (m < j ? m : m - 1)) * N;
        const u64 *k0 = key + (((size_t)j * 2 + 0) * K + pm) * N;
        const u64 *k1 = key + (((size_t)j * 2 + 1) * K + pm) * N;
        const u64x2 x = *reinterpret_cast<const u64x2 *>(op + k);
        const u64x2 y0 = *reinterpret_cast<const u64x2 *>(k0 + k);
        const u64x2 y1 = *reinterpret_cast<const u64x2 *>(k1 + k);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            a0[e].mac(x[e], y0[e]);
            a1[e].mac(x[e], y1[e]);
        }
        if ((j & 15) == 15) { // keep the 128-bit sums below 2^124
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
    *reinterpret_cast<u64x2 *>(acc + ((size_t)0 * (ell + 1) + m) * N + k) = o0;
    *reinterpret_cast<u64x2 *>(acc + ((size_t)1 * (ell + 1) + m) * N + k) = o1;
}

// divide-and-round, first half (RNSTool::divide_and_round_q_last_ntt_inplace / the mod-down of switch_key):
// last[p] is the dropped limb in the coefficient domain, canonical mod q_l.
//   tmp[p][i] = ((last + floor(q_l/2)) mod q_l) mod q_i  -  floor(q_l/2) mod q_i        (mod q_i), i < cnt
// grid = (N/512, cnt, polys)
__global__ __launch_bounds__(kOpThreads) void dr_lift_kernel(u64 *__restrict__ tmp, long tmp_poly_stride,
                                                              const u64 *__restrict__ last, long last_poly_stride, int l,
                                                              int K, size_t N, const DModulus *__restrict__ mods,
                                                              const u64 *__restrict__ half_mod)
{
    const int i = blockIdx.y, p = blockIdx.z;
    const DModulus Mi = mods[i];
    const u64 ql = mods[l].q, qi = Mi.q, half = ql >> 1;
    const u64 neg_half = qi - half_mod[(size_t)l * K + i]; // in (0, qi]
    const size_t k = ((size_t)blockIdx.x * kOpThreads + threadIdx.x) * 2;
    const u64x2 v = *reinterpret_cast<const u64x2 *>(last + p * last_poly_stride + k);
    u64x2 r;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        u64 y = v[e] + half;
        y = y >= ql ? y - ql : y;
        y = recanon(y, Mi);
        y += neg_half;
        r[e] = y >= qi ? y - qi : y;
    }
    *reinterpret_cast<u64x2 *>(tmp + p * tmp_poly_stride + (size_t)i * N + k) = r;
}

// divide-and-round, second half: out[p][i] = base[p][i] + (x[p][i] - tmp[p][i]) * q_l^{-1}  (mod q_i)
__global__ __launch_bounds__(kOpThreads) void dr_final_kernel(CtView out, const u64 *__restrict__ x, long x_poly_stride,
                                                               const u64 *__restrict__ tmp, long tmp_poly_stride,
                                                               const u64 *base0, const u64 *base1, int l, int K, size_t N,
                                                               const DModulus *__restrict__ mods,
                                                               const u64 *__restrict__ inv_last)
{
    const int i = blockIdx.y, p = blockIdx.z;
    const DModulus M = mods[i];
    const u64 inv = inv_last[(size_t)l * K + i];
    const u64 *base = p ? base1 : base0;
    const size_t k = ((size_t)blockIdx.x * kOpThreads + threadIdx.x) * 2;
    const u64x2 xv = *reinterpret_cast<const u64x2 *>(x + p * x_poly_stride + (size_t)i * N + k);
    const u64x2 tv = *reinterpret_cast<const u64x2 *>(tmp + p * tmp_poly_stride + (size_t)i * N + k);
    u64x2 bv = { 0, 0 };
    if (base) bv = *reinterpret_cast<const u64x2 *>(base + (size_t)i * N + k);
    u64x2 r;
#pragma unroll
    for (int e = 0; e < 2; e++) r[e] = addmod(bv[e], mulmod(submod(xv[e], tv[e], M.q), inv, M), M.q);
    *reinterpret_cast<u64x2 *>(out.limb(p, i, N) + k) = r;
}

__global__ __launch_bounds__(kOpThreads) void copy_limbs_kernel(u64 *__restrict__ dst, long dst_stride,
                                                                 const u64 *__restrict__ src, long src_stride)
{
    const size_t k = ((size_t)blockIdx.x * kOpThreads + threadIdx.x) * 2;
    *reinterpret_cast<u64x2 *>(dst + blockIdx.y * dst_stride + k) =
        *reinterpret_cast<const u64x2 *>(src + blockIdx.y * src_stride + k);
}

void keyswitch(Context &c, const Workspace &w, CtView out, const u64 *base0, const u64 *base1, const u64 *target,
               const u64 *key, int ell, hipStream_t s)
{
    if (c.hybrid()) {
        hyb_keyswitch(c, w, out, base0, base1, target, key, ell, s);
        return;
    }
    const size_t N = c.N;
    const int K = c.K, sp = K - 1;
    const unsigned gx = (unsigned)(N / (2 * kOpThreads));
    // (1) digits = iNTT(target)
    DC_LAUNCH(copy_limbs_kernel, dim3(gx, ell), dim3(kOpThreads), 0, s, w.ks_digits, (long)N, target, (long)N);
    launch_ntt(c, true, w.ks_digits, (long)N, ell, nullptr, 0, 0, s);
    // (2) lift every digit to every other modulus, forward NTT there
    DC_LAUNCH(ks_lift_kernel, dim3(gx, ell, ell), dim3(kOpThreads), 0, s, w.ks_ext, w.ks_digits, ell, sp, N,
                       c.d_mods);
    launch_ntt(c, false, w.ks_ext, (long)N, ell * ell, c.ks_prime_idx(ell), 0, 0, s);
    // (3) inner products with the key
    DC_LAUNCH(ks_mac_kernel, dim3(gx, ell + 1), dim3(kOpThreads), 0, s, w.ks_acc, w.ks_ext, target, key, ell, K, N,
                       c.d_mods);
    // (4) mod-down by the special prime
    u64 *acc_last = w.ks_acc + (size_t)ell * N;
    const long acc_ps = (long)(ell + 1) * (long)N;
    launch_ntt(c, true, acc_last, acc_ps, 2, nullptr, sp, 1, s);
    DC_LAUNCH(dr_lift_kernel, dim3(gx, ell, 2), dim3(kOpThreads), 0, s, w.ks_tmp, (long)ell * (long)N, acc_last,
                       acc_ps, sp, K, N, c.d_mods, c.d_half_mod);
    launch_ntt(c, false, w.ks_tmp, (long)N, 2 * ell, nullptr, 0, ell, s);
    DC_LAUNCH(dr_final_kernel, dim3(gx, ell, 2), dim3(kOpThreads), 0, s, out, w.ks_acc, acc_ps, w.ks_tmp,
                       (long)ell * (long)N, base0, base1, sp, K, N, c.d_mods, c.d_inv_last);
}

void rescale(Context &c, const Workspace &w, CtView dst, CtView src, int ell, hipStream_t s)
{
    const size_t N = c.N;
    const int l = ell - 1;
    const unsigned gx = (unsigned)(N / (2 * kOpThreads));
    u64 *last = w.ks_digits; // [2][N]
    DC_LAUNCH(copy_limbs_kernel, dim3(gx, 2), dim3(kOpThreads), 0, s, last, (long)N, src.limb(0, l, N),
                       src.poly_stride);
    launch_ntt(c, true, last, (long)N, 2, nullptr, l, 1, s);
    if (l == 0) return;
    DC_LAUNCH(dr_lift_kernel, dim3(gx, l, 2), dim3(kOpThreads), 0, s, w.ks_tmp, (long)l * (long)N, last, (long)N, l,
                       c.K, N, c.d_mods, c.d_half_mod);
    launch_ntt(c, false, w.ks_tmp, (long)N, 2 * l, nullptr, 0, l, s);
    DC_LAUNCH(dr_final_kernel, dim3(gx, l, 2), dim3(kOpThreads), 0, s, dst, src.p, src.poly_stride, w.ks_tmp,
                       (long)l * (long)N, (const u64 *)nullptr, (const u64 *)nullptr, l, c.K, N, c.d_mods, c.d_inv_last);
}

void mul_relin(Context &c, const Workspace &w, CtView dst, CtView a, CtView b, const u64 *relin_key, int ell, hipStream_t s)
{
    u64 *c2 = w.ct_tmp;
    launch_tensor(c, dst, c2, a, b, ell, s);
    keyswitch(c, w, dst, dst.limb(0, 0, c.N), dst.limb(1, 0, c.N), c2, relin_key, ell, s);
}

void rotate_hop(Context &c, const Workspace &w, CtView dst, CtView src, u32 galois_elt, const u64 *galois_key, int ell,
                hipStream_t s)
{
    if (c.hybrid()) { // grouped-digit mode decomposes before the automorphism (hybrid_ks.hip): its own single-hop sequence, source kept apart from dst
        CtView from = src;
        if (src.p == dst.p) {
            from = CtView{ w.ct_tmp, (long)ell * (long)c.N };
            launch_ew(c, EwOp::Copy, from, src, src, 2, 2, ell, s);
        }
        hyb_rotate_hop_single(c, w, dst, from, galois_elt, galois_key, ell, s);
        return;
    }
    // permuted (c0, c1) -> scratch [2][ell][N]; c1' is the key-switch target, c0' the base of output poly 0
    CtView tmp{ w.ct_tmp, (long)ell * (long)c.N };
    launch_galois(c, tmp, src, galois_elt, 2, ell, s);
    keyswitch(c, w, dst, tmp.limb(0, 0, c.N), nullptr, tmp.limb(1, 0, c.N), galois_key, ell, s);
}

} // namespace dacapo
