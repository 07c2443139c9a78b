// Limb-wise kernels of the HEVM path: the single-pass opcodes (negate/addcc/addcp/mulcp, SEAL_HEVM.cpp:275-323),
// the ckks tensor product (:315), the NTT-domain Galois permutation (:273) and the glue passes of key switching
// and rescaling.  All are streaming kernels: 16 B per lane, grid = (N/512, limbs, polys).
#include "kernels.hpp"

namespace dacapo {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
constexpr int kEwThreads = 256;

template <int OP>
__device__ __forceinline__ u64 ew_apply(u64 a, u64 b, const DModulus &m)
{
    if (OP == (int)EwOp::Add) return addmod(a, b, m.q);
    if (OP == (int)EwOp::Sub) return submod(a, b, m.q);
    if (OP == (int)EwOp::Neg) return negmod(a, m.q);
    if (OP == (int)EwOp::Mul) return mulmod(a, b, m);
    return a;
}

template <int OP>
__global__ __launch_bounds__(kEwThreads) void ew_kernel(CtView dst, CtView a, CtView b, int b_polys, size_t N,
                                                         const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, p = blockIdx.z;
    const DModulus m = mods[i];
    const size_t k = ((size_t)blockIdx.x * kEwThreads + threadIdx.x) * 2;
    const u64x2 va = *reinterpret_cast<const u64x2 *>(a.limb(p, i, N) + k);
    u64x2 vb = va;
    if (OP != (int)EwOp::Neg && OP != (int)EwOp::Copy)
        vb = *reinterpret_cast<const u64x2 *>(b.limb(b_polys == 1 ? 0 : p, i, N) + k);
    u64x2 r;
    r.x = ew_apply<OP>(va.x, vb.x, m);
    r.y = ew_apply<OP>(va.y, vb.y, m);
    *reinterpret_cast<u64x2 *>(dst.limb(p, i, N) + k) = r;
}

void launch_ew(const Context &c, EwOp op, CtView dst, CtView a, CtView b, int polys, int b_polys, int ell, hipStream_t s)
{
    dim3 grid((unsigned)(c.N / (2 * kEwThreads)), (unsigned)ell, (unsigned)polys), block(kEwThreads);
    switch (op) {
    case EwOp::Add: DC_LAUNCH(ew_kernel<0>, grid, block, 0, s, dst, a, b, b_polys, c.N, c.d_mods); break;
    case EwOp::Sub: DC_LAUNCH(ew_kernel<1>, grid, block, 0, s, dst, a, b, b_polys, c.N, c.d_mods); break;
    case EwOp::Neg: DC_LAUNCH(ew_kernel<2>, grid, block, 0, s, dst, a, b, b_polys, c.N, c.d_mods); break;
    case EwOp::Mul: DC_LAUNCH(ew_kernel<3>, grid, block, 0, s, dst, a, b, b_polys, c.N, c.d_mods); break;
    case EwOp::Copy: DC_LAUNCH(ew_kernel<4>, grid, block, 0, s, dst, a, b, b_polys, c.N, c.d_mods); break;
    }
}

__global__ __launch_bounds__(kEwThreads) void add_plain_kernel(CtView dst, CtView a, const u64 *__restrict__ pt, size_t N,
                                                                const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y, p = blockIdx.z;
    const size_t k = ((size_t)blockIdx.x * kEwThreads + threadIdx.x) * 2;
    u64x2 v = *reinterpret_cast<const u64x2 *>(a.limb(p, i, N) + k);
    if (p == 0) {
        const u64 q = mods[i].q;
        const u64x2 w = *reinterpret_cast<const u64x2 *>(pt + (size_t)i * N + k);
        v.x = addmod(v.x, w.x, q);
        v.y = addmod(v.y, w.y, q);
    }
    *reinterpret_cast<u64x2 *>(dst.limb(p, i, N) + k) = v;
}

void launch_add_plain(const Context &c, CtView dst, CtView a, const u64 *pt, int ell, hipStream_t s)
{
    // c1 only moves when dst is a different register
    const int polys = (dst.p == a.p && dst.poly_stride == a.poly_stride) ? 1 : 2;
    dim3 grid((unsigned)(c.N / (2 * kEwThreads)), (unsigned)ell, (unsigned)polys);
    DC_LAUNCH(add_plain_kernel, grid, dim3(kEwThreads), 0, s, dst, a, pt, c.N, c.d_mods);
}

__global__ __launch_bounds__(kEwThreads) void tensor_kernel(CtView dst, u64 *__restrict__ c2out, CtView a, CtView b,
                                                             size_t N, const DModulus *__restrict__ mods)
{
    const int i = blockIdx.y;
    const DModulus m = mods[i];
    const size_t k = ((size_t)blockIdx.x * kEwThreads + threadIdx.x) * 2;
    const u64x2 a0 = *reinterpret_cast<const u64x2 *>(a.limb(0, i, N) + k);
    const u64x2 a1 = *reinterpret_cast<const u64x2 *>(a.limb(1, i, N) + k);
    const u64x2 b0 = *reinterpret_cast<const u64x2 *>(b.limb(0, i, N) + k);
    const u64x2 b1 = *reinterpret_cast<const u64x2 *>(b.limb(1, i, N) + k);
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
    *reinterpret_cast<u64x2 *>(dst.limb(0, i, N) + k) = c0;
    *reinterpret_cast<u64x2 *>(dst.limb(1, i, N) + k) = c1;
    *reinterpret_cast<u64x2 *>(c2out + (size_t)i * N + k) = c2;
}

void launch_tensor(const Context &c, CtView dst, u64 *c2out, CtView a, CtView b, int ell, hipStream_t s)
{
    dim3 grid((unsigned)(c.N / (2 * kEwThreads)), (unsigned)ell);
    DC_LAUNCH(tensor_kernel, grid, dim3(kEwThreads), 0, s, dst, c2out, a, b, c.N, c.d_mods);
}

// GaloisTool::apply_galois_ntt: out[k] = in[bitrev(((elt * (2*bitrev(k)+1)) >> 1) mod N)].  An aligned block of
// 2^b consecutive k reads an aligned block of 2^b consecutive inputs (elt is odd), so the gather stays
// coalesced at 64-lane granularity; no permutation table is needed (v_bfrev_b32 does the bit reversals).
__device__ __forceinline__ u32 galois_src_index(u32 k, u32 elt, int logN)
{
    const u32 r = (__brev(k) >> (32 - logN)) * 2u + 1u;
    const u32 idx = ((elt * r) >> 1) & ((1u << logN) - 1u);
    return __brev(idx) >> (32 - logN);
}

__global__ __launch_bounds__(kEwThreads) void galois_kernel(CtView dst, CtView src, u32 elt, int logN)
{
    const size_t N = (size_t)1 << logN;
    const int i = blockIdx.y, p = blockIdx.z;
    const u32 k = blockIdx.x * kEwThreads + threadIdx.x;
    dst.limb(p, i, N)[k] = src.limb(p, i, N)[galois_src_index(k, elt, logN)];
}

void launch_galois(const Context &c, CtView dst, CtView src, u32 galois_elt, int polys, int ell, hipStream_t s)
{
    dim3 grid((unsigned)(c.N / kEwThreads), (unsigned)ell, (unsigned)polys);
    DC_LAUNCH(galois_kernel, grid, dim3(kEwThreads), 0, s, dst, src, galois_elt, c.logN);
}

} // namespace dacapo
