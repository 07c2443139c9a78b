// CKKS encoder on the device, batched: what preprocess() (SEAL_HEVM.cpp:242-267) does once per opcode-0 instruction --
// CKKSEncoder::encode [SEAL-upstream ckks.h encode_internal: slot permutation, inverse "special" FFT over C^N, scale, round]
// followed by the reduction into the level's primes and the forward NTT -- for all plaintext registers of a program in a
// few launches.  The arithmetic is HostEncoder::encode's operation for operation (same butterfly order, same root table,
// same complex-multiply formula, IEEE double with contraction disabled), so the coefficients are bit-identical to the
// host encoder's; only who executes them changes (5 894 plaintexts of ResNet-20: 3 s of host FFTs -> ~0.1 s).
#pragma clang fp contract(off)
#include "encoder.hpp"

namespace dacapo {

constexpr int kEncThreads = 256;

// v[p][slot_map[i]] = v[p][slot_map[slots + i]] = src[i % len]  (a real value is its own conjugate).  grid = (slots/256, P)
__global__ __launch_bounds__(kEncThreads) void enc_scatter_kernel(double2 *__restrict__ v, const double *__restrict__ consts,
                                                                   const EncItem *__restrict__ items, const u32 *__restrict__ slot_map,
                                                                   size_t N)
{
    const size_t slots = N >> 1, i = (size_t)blockIdx.x * kEncThreads + threadIdx.x;
    const EncItem it = items[blockIdx.y];
    const double x = it.len ? consts[it.src_off + i % it.len] : 1.0; // len 0: the all-ones "upscale" constant
    const double y = it.cplx ? consts[it.src_off + it.len + i % it.len] : 0.0;
    double2 *o = v + (size_t)blockIdx.y * N;
    o[slot_map[i]] = make_double2(x, y);
    o[slot_map[slots | i]] = make_double2(x, -y); // the conjugate half (CKKSEncoder::encode_internal)
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// one Gentleman-Sande stage of transform_from_rev: group i of m, butterfly j of gap; r = conj(root[m + i]).
// grid = (N/2/256, P).  last = the m == 1 stage with the scalar scale/N merged (per item).
__global__ __launch_bounds__(kEncThreads) void enc_stage_kernel(double2 *__restrict__ v, const double2 *__restrict__ roots,
                                                                 const EncItem *__restrict__ items, size_t N, unsigned log_gap, int last)
{
    const size_t t = (size_t)blockIdx.x * kEncThreads + threadIdx.x;
    const size_t gap = (size_t)1 << log_gap, i = t >> log_gap, j = t & (gap - 1), m = (N >> 1) >> log_gap;
    double2 *x = v + (size_t)blockIdx.y * N + 2 * i * gap + j, *y = x + gap;
    const double2 a = *x, b = *y;
    const double2 rt = roots[m + i];
    double2 r = make_double2(rt.x, -rt.y);
    const double2 s = make_double2(a.x + b.x, a.y + b.y), d = make_double2(a.x - b.x, a.y - b.y);
    if (!last) {
        *x = s;
        *y = cmul(d, r);
    } else {
        const double fix = items[blockIdx.y].fix;
        r = make_double2(r.x * fix, r.y * fix);
        *x = make_double2(s.x * fix, s.y * fix);
        *y = cmul(d, r);
    }
}

// round the real parts, reduce into the first `level` primes: out[p][k][j].  grid = (N/256, P)
// (prime_base: the limbs are those of primes prime_base ... prime_base + level - 1 -- 0 everywhere but the special-prime limbs of double hoisting)
__global__ __launch_bounds__(kEncThreads) void enc_round_lift_kernel(u64 *__restrict__ out, const double2 *__restrict__ v, int level, size_t N,
                                                                      const DModulus *__restrict__ mods, int *__restrict__ overflow, int prime_base)
{
    const size_t j = (size_t)blockIdx.x * kEncThreads + threadIdx.x;
    const double c = round(v[(size_t)blockIdx.y * N + j].x);
    const double a = fabs(c);
    if (!(a < 0x1p120)) {
        *overflow = 1;
        return;
    }
    const bool neg = c < 0.0;
    u64 l, h;
    if (a < 0x1p63) {
        l = (u64)a;
        h = 0;
    } else { // exact: a is integral and has at most 53 significant bits
        int e;
        const double fr = frexp(a, &e);
        const u64 mant = (u64)ldexp(fr, 53);
        const int sh = e - 53;
        l = sh < 64 ? mant << sh : 0;
        h = sh < 64 ? mant >> (64 - sh) : mant << (sh - 64);
    }
    u64 *o = out + (size_t)blockIdx.y * level * N + j;
    for (int k = 0; k < level; k++) {
        const DModulus M = mods[prime_base + k];
        const u64 r = reduce128_any(h, l, M);
        o[(size_t)k * N] = (neg && r) ? M.q - r : r;
    }
}

// ---- decoder: CKKSEncoder::decode's forward special FFT (transform_to_rev: Cooley-Tukey butterflies, root[m + i]) -------------
// v[N] holds the plaintext coefficients already composed and divided by the scale (real parts; see dec_crt_kernel in
// hevm_vm.hip).  One launch per stage: group i of m, butterfly j of gap.  grid = (N/2/256)
__global__ __launch_bounds__(kEncThreads) void dec_stage_kernel(double2 *__restrict__ v, const double2 *__restrict__ roots, size_t N,
                                                                 unsigned log_gap)
{
    const size_t t = (size_t)blockIdx.x * kEncThreads + threadIdx.x;
    const size_t gap = (size_t)1 << log_gap, i = t >> log_gap, j = t & (gap - 1), m = (N >> 1) >> log_gap;
    double2 *x = v + 2 * i * gap + j, *y = x + gap;
    const double2 a = *x, b = cmul(*y, roots[m + i]);
    *x = make_double2(a.x + b.x, a.y + b.y);
    *y = make_double2(a.x - b.x, a.y - b.y);
}

// out[i] = Re v[slot_map[i]] for the N/2 slots.  grid = (N/2/256)
__global__ __launch_bounds__(kEncThreads) void dec_gather_kernel(double *__restrict__ out, const double2 *__restrict__ v,
                                                                  const u32 *__restrict__ slot_map)
{
    const size_t i = (size_t)blockIdx.x * kEncThreads + threadIdx.x;
    out[i] = v[slot_map[i]].x;
}

void dec_fft(const Context &c, const EncTables &tb, double2 *v, double *out, hipStream_t s)
{
    const size_t N = c.N;
    for (int lg = c.logN - 1; lg >= 0; lg--) // m = 1, 2, ..., N/2  <=>  gap = N/2, ..., 1
        DC_LAUNCH(dec_stage_kernel, dim3((unsigned)(N / 2 / kEncThreads)), dim3(kEncThreads), 0, s, v, tb.roots, N, (unsigned)lg);
    DC_LAUNCH(dec_gather_kernel, dim3((unsigned)(N / 2 / kEncThreads)), dim3(kEncThreads), 0, s, out, v, tb.slot_map);
}

void enc_batch(const Context &c, const EncTables &tb, const double *d_consts, const EncItem *d_items, int P, int level, double2 *scratch,
               u64 *out, int *d_overflow, hipStream_t s, int prime_base)
{
    const size_t N = c.N;
    DC_LAUNCH(enc_scatter_kernel, dim3((unsigned)(N / 2 / kEncThreads), (unsigned)P), dim3(kEncThreads), 0, s, scratch, d_consts, d_items,
                       tb.slot_map, N);
    for (int lg = 0; lg < c.logN; lg++) // gap = 1, 2, ..., N/2  <=>  m = N/2, ..., 1
        DC_LAUNCH(enc_stage_kernel, dim3((unsigned)(N / 2 / kEncThreads), (unsigned)P), dim3(kEncThreads), 0, s, scratch, tb.roots, d_items,
                           N, (unsigned)lg, lg == c.logN - 1 ? 1 : 0);
    DC_LAUNCH(enc_round_lift_kernel, dim3((unsigned)(N / kEncThreads), (unsigned)P), dim3(kEncThreads), 0, s, out, scratch, level, N,
                       c.d_mods, d_overflow, prime_base);
    launch_ntt(c, false, out, (long)N, P * level, nullptr, prime_base, level, s);
}

} // namespace dacapo
