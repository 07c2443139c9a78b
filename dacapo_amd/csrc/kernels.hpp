// Launchers for the HIP kernels of the HEVM hot path (definitions: ntt_kernels.hip, poly_kernels.hip,
// keyswitch_kernels.hip).  Everything is stream-ordered; no launcher allocates, copies or synchronises.
#pragma once
#include "context.hpp"

namespace dacapo {

// Polynomial data in HBM is always uint64 limbs, limb-major, N coefficients per limb, NTT domain unless
// stated.  A ciphertext register is [poly][limb][N] with a fixed poly stride (capacity), so dropping a limb
// (modswitch) never moves data.
struct CtView {
    u64 *p;
    long poly_stride; // in elements
    __host__ __device__ u64 *limb(int poly, int i, size_t N) const { return p + poly * poly_stride + (long)i * (long)N; }
};

// ---- NTT (ntt_kernels.hip) -----------------------------------------------------------------------------
// `count` limbs at data + b*limb_stride; limb b is modulo prime d_prime_idx[b % prime_period] (device array) or,
// when the pointer is null, prime_base + (b % prime_period); prime_period <= 0 means "no wrap".  In place.
// Output canonical.
void launch_ntt(const Context &c, bool inverse, u64 *data, long limb_stride, int count, const int *d_prime_idx,
                int prime_base, int prime_period, hipStream_t s);

// the two-launch transform whatever the batch size
void launch_ntt_two_phase(const Context &c, bool inverse, u64 *data, long limb_stride, int count, const int *d_prime_idx,
                          int prime_base, int prime_period, hipStream_t s);
// single-crossing transform (ntt_full.hip): one 1024-thread workgroup per limb, N = 2^15 only.  launch_ntt takes it for launches of
// at least ntt_full_min_limbs() limbs (option ntt_full_min_limbs / option ntt_full_inv_min_limbs; 0 = never)
bool ntt_full_supported(const Context &c);
long ntt_full_min_limbs(bool inverse);
bool ntt_full_pays(bool inverse, int count); // ... and the launch fills the persistent grid's last round well enough (ntt_full.hip)
void launch_ntt_full(const Context &c, bool inverse, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                     int prime_period, hipStream_t s);

// second (ROWS) phase of a forward NTT only: the input holds the output of a COLS phase (fused_ks.hip)
void launch_ntt_rows_fwd(const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                         int prime_period, hipStream_t s);

// first (COLS) phase of a forward NTT only (lazy output; a fused ROWS-phase kernel finishes the transform)
void launch_ntt_cols_fwd(const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                         int prime_period, hipStream_t s);

// second (COLS) phase of an inverse NTT only: the input holds the output of an inverse ROWS phase
// `mods`: another table of per-prime constants (indexed like Context::d_mods) -- the fused grouped-digit key switch passes copies whose
// N^-1 words carry a base conversion's per-input constant, so that the transform's last stage applies it for free
void launch_ntt_cols_inv(const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                         int prime_period, hipStream_t s, const DModulus *mods = nullptr);
// first (ROWS) phase of an inverse NTT only (lazy output)
void launch_ntt_rows_inv(const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                         int prime_period, hipStream_t s);

// ---- limb-wise kernels (poly_kernels.hip) -----------------------------------------------------------------
enum class EwOp : int { Add = 0, Sub = 1, Neg = 2, Mul = 3, Copy = 4 };
// dst[p][i] = a[p][i] (op) b[pb][i] for p < polys, i < ell (limb i modulo prime i).  b_polys == 1 broadcasts
// a single polynomial (plaintext) to every poly of a.  dst may alias a and/or b.
void launch_ew(const Context &c, EwOp op, CtView dst, CtView a, CtView b, int polys, int b_polys, int ell, hipStream_t s);
// addcp: dst0 = a0 + pt, dst1 = a1
void launch_add_plain(const Context &c, CtView dst, CtView a, const u64 *pt, int ell, hipStream_t s);
// ckks_multiply: (c0,c1,c2) = (a0 b0, a0 b1 + a1 b0, a1 b1); c0,c1 -> dst, c2 -> c2out[ell][N].  dst may alias a/b.
void launch_tensor(const Context &c, CtView dst, u64 *c2out, CtView a, CtView b, int ell, hipStream_t s);
// apply_galois_ntt on `polys` polynomials: dst[p][i][k] = src[p][i][perm(k)] (dst must not alias src)
void launch_galois(const Context &c, CtView dst, CtView src, u32 galois_elt, int polys, int ell, hipStream_t s);

// ---- composite ops (ckks_ops.hip) ---------------------------------------------------------------------------
// Evaluator::switch_key_inplace: (out0,out1) (+)= KS(target) at level ell.  key: [K-1][2][K][N].
// base0/base1: what the switched pair is added to (nullable = 0); out may alias base.  target [ell][N] NTT form
// (preserved).
void keyswitch(Context &c, const Workspace &w, CtView out, const u64 *base0, const u64 *base1, const u64 *target,
               const u64 *key, int ell, hipStream_t s);
// Evaluator::rescale_to_next: dst(level ell-1) = round(src / q_{ell-1}).  dst may alias src.
void rescale(Context &c, const Workspace &w, CtView dst, CtView src, int ell, hipStream_t s);
// Evaluator::multiply + relinearize_inplace
void mul_relin(Context &c, const Workspace &w, CtView dst, CtView a, CtView b, const u64 *relin_key, int ell, hipStream_t s);
// Evaluator::apply_galois_inplace (one key-switch hop)
void rotate_hop(Context &c, const Workspace &w, CtView dst, CtView src, u32 galois_elt, const u64 *galois_key, int ell,
                hipStream_t s);

} // namespace dacapo
