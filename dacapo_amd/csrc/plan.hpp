// Batched execution plan of an HEVM program.
//
// The reference interprets the bytecode one blocking SEAL call at a time (SEAL_HEVM.cpp:336-401).  On the MI355X a
// single low-level ciphertext op (2-5 primes) is a few microseconds of work spread over a dozen dependent launches:
// issued one by one the GPU idles between them.  HBM is plentiful (288 GB), so run() instead executes a PLAN
// built once per loaded program:
//   1. registers are renamed (SSA): every op result gets its own buffer from a pool, which removes the WAR/WAW
//      hazards that ReuseBuffer.cpp's register recycling introduces and leaves only true dataflow;
//   2. ops are levelled by dataflow depth ("waves"); all ops of one kind and level in a wave become ONE batched
//      launch sequence (the key switches of the ~100 independent rotations of a convolution run as one batch);
//   3. chains of ct+ct additions are summed by one n-ary kernel -- modular addition is exact, so the result limbs
//      are bit-identical to the sequential chain.
// Every kernel takes a device-resident table of per-item views; level/scale bookkeeping is resolved at plan time.
#pragma once
#include "crt_device.hpp"
#include "kernels.hpp"

namespace dacapo {

struct KsItem {   // one key-switch hop of a rotation: dst = apply_galois(src)
    CtView src, dst;
    const u64 *key;
    u32 elt;
    u32 slot;     // grouped-digit mode: which decomposition of the batch this hop reads (hops of one source ciphertext share one); else 0
    // double hoisting (option hyb_double_hoist; items of a lazy sum only): the plaintext that multiplies this rotation's result before it joins
    // the sum -- its limbs over the data primes [level][N] and over the special primes [ksp][N] (both NTT form) -- or null
    const u64 *plain = nullptr, *plain_sp = nullptr;
};
struct MulItem {  // dst = relinearize(a * b)
    CtView a, b, dst;
};
struct RsItem {   // dst = rescale_to_next(expr), expr = ((src | sum of `count` terms srcs[first ...]) + add on c0) * mul
    CtView src, dst;
    int first = 0, count = 0;          // count == 0: expr starts from src
    const u64 *add = nullptr;          // plaintext [level][N] added to c0 (a single-use addcp folded in), or null
    const u64 *mul = nullptr;          // plaintext multiplying both polys (a single-use mulcp folded in), or null
};
struct EwItem {   // dst = a (op) b ; plaintext operand: b.p = limbs, b.poly_stride = 0
    CtView dst, a, b;
};
struct SumSrc {   // one term of a sum: a ciphertext, optionally times a plaintext (a ct*pt product folded into the sum)
    CtView v;
    const u64 *plain; // [level][N] or null
};
struct SumItem {  // dst = sum of `count` terms srcs[first ...]
    CtView dst;
    int first, count;
};

// n-ary sums of one step that name the same ciphertexts, run together (batch_ops.hip b_sum_group_kernel): up to kSumGroup items to a thread,
// the union of their sources walked once -- a rotated ciphertext that feeds 8 output channels of a convolution, or 8 giant steps of a
// matrix-vector product, is read once instead of 8 times.
constexpr int kSumGroup = 8;
struct SumGroupSrc { // one source of a group and, per item, what multiplies it
    CtView v;
    const u64 *plain[kSumGroup]; // [level][N]; null: the item does not use this source; kSumAddOnly: the item adds it as it is
};
struct SumGroup {
    CtView dst[kSumGroup];
    int first, count, items; // sources gsrcs[first ... first + count), items <= kSumGroup
};
__host__ __device__ inline const u64 *sum_add_only() { return reinterpret_cast<const u64 *>((uintptr_t)8); }

struct BootItem { // opcode 10: dst = Enc(re-encode(Dec(src))) = zenc + (plaintext, 0)
    CtView src, dst;
    const u64 *zenc; // fresh public-key encryption of zero at the target level, [2][t][N], made at the start of the run
    double ratio;    // new_scale / old_scale (divided by the dropped prime when a rescale has been folded in, see plan_exec.hip)
    // operand expression like RsItem's: ((src | sum of `count` terms) + add on c0) * mul, decrypted in the first loader
    int first = 0, count = 0;
    const u64 *add = nullptr, *mul = nullptr;
};

// A step's link to a fused neighbour.  The last kernel of every composite op is a forward ROWS phase and the first kernel of
// every composite op an inverse ROWS phase over the SAME tiles (ntt_tile.hpp: the last pass of one and the first pass of the
// other use the same thread <-> coefficient map).  When the plan sees that step B only consumes what step A produces, A's last
// kernel keeps going: it stores its result, forms B's first-phase operand from the values still in registers and runs B's
// inverse ROWS phase, writing B's first-phase buffer -- one launch (and one trip through HBM) fewer per link.
enum ContKind : int {
    CONT_NONE = 0,
    CONT_RS = 1,   // consumer is a rescale: inverse phase of the limb it drops (both polynomials)
    CONT_MUL = 2,  // consumer is a ct x ct multiply: inverse phase of out.c1 * other.c1 (the tensor product's c2), every limb
    CONT_BOOT = 3, // consumer is an opcode 10: inverse phase of out.c0 + out.c1 * s (Decryptor::decrypt), every limb
};
struct Handoff {
    int cont = CONT_NONE;          // what this step's last kernel also computes ...
    u64 *out = nullptr;            // ... into the consumer's first-phase buffer ([B][2][N], [B][l][N], [B][l][N])
    const CtView *other = nullptr; // CONT_MUL: device table [B] of the consumer's other operand (p == nullptr: a square)
    const struct RsItem *rs_items = nullptr; // CONT_RS: the consumer's items (a "+ plaintext" / "* plaintext" folded into it applies first)
    const u64 *sk = nullptr;       // CONT_BOOT: secret key [K][N]
    const u64 *in = nullptr;       // this step's own first phase was computed by its producer: read it here instead of launching it
};

// scratch of one batched step, sized for the largest batch of the plan
struct BatchWs {
    u64 *target = nullptr; // [B][l][N]     key-switch target, NTT form
    u64 *digits = nullptr; // [B][l][N]     its coefficient-domain digits (rescale: [B][2][N] dropped limbs)
    u64 *ext = nullptr;    // [B][l*l][N]   digits lifted to the other moduli
    u64 *acc = nullptr;    // [B][2][l+1][N]
    u64 *tmp = nullptr;    // [B][2][l][N]
};

// `unique`: grouped-digit mode only -- the number of distinct decompositions the items' `slot` fields name (0: every item its own)
void b_rotate_hops(Context &c, const BatchWs &w, const KsItem *d_items, int B, int ell, hipStream_t s, const Handoff &h = Handoff{}, int unique = 0);
void b_mul_relin(Context &c, const BatchWs &w, const MulItem *d_items, const u64 *relin_key, int B, int ell, hipStream_t s,
                 const Handoff &h = Handoff{});
void b_rescale(Context &c, const BatchWs &w, const RsItem *d_items, int B, int ell, hipStream_t s, const SumSrc *d_srcs = nullptr,
               const Handoff &h = Handoff{});
bool chain_fusion_supported(); // the continuation kernels exist for the default launch sequences only
// grouped-digit hybrid key switching (hybrid_ks.hip; Context::hybrid()): b_rotate_hops / b_mul_relin / keyswitch route here
void hyb_rotate_hops(Context &c, const BatchWs &w, const KsItem *d_items, int B, int ell, hipStream_t s, int unique = 0);
// lazy sums (option hyb_lazy_sum): the B items' accumulators are added group by group and divided by P once per group.  d_groups[g]: dst = the
// sum's destination, elt = first item of the group, slot = its item count (a group's items are adjacent in d_items; their own dst is unused)
void hyb_rotate_sum(Context &c, const BatchWs &w, const KsItem *d_items, int B, const KsItem *d_groups, int G, int ell, hipStream_t s, int unique = 0);
bool hyb_lazy_sum_supported(const Context &c);
// one rotation hop by value (the one-instruction-at-a-time loop): src must not alias dst
void hyb_rotate_hop_single(Context &c, const Workspace &w, CtView dst, CtView src, u32 galois_elt, const u64 *galois_key, int ell, hipStream_t s);
void hyb_mul_relin(Context &c, const BatchWs &w, const MulItem *d_items, const u64 *relin_key, int B, int ell, hipStream_t s);
void hyb_keyswitch(Context &c, const Workspace &w, CtView out, const u64 *base0, const u64 *base1, const u64 *target, const u64 *key, int ell,
                   hipStream_t s);
// NTT-equivalents of one key switch at level ell: (l + 1)(l + 2) for SEAL's scheme, G (l + ksp) + 2 ksp + 2 l with grouped digits
inline int64_t ks_ntt_count(const Context &c, int ell)
{
    return c.hybrid() ? (int64_t)c.hyb_groups(ell) * (ell + c.ksp) + 2 * c.ksp + 2 * ell : (int64_t)(ell + 1) * (ell + 2);
}
// op: Neg / Mul (ct * plain) / Copy ; b_polys as in launch_ew
void b_ew(Context &c, EwOp op, const EwItem *d_items, int B, int polys, int b_polys, int ell, hipStream_t s);
void b_add_plain(Context &c, const EwItem *d_items, int B, int ell, hipStream_t s);
void b_sum(Context &c, const SumItem *d_items, const SumSrc *d_srcs, int B, int ell, hipStream_t s);
void b_sum_group(Context &c, const SumGroup *d_groups, const SumGroupSrc *d_gsrcs, int G, int ell, hipStream_t s);
// ModRaise (extension opcode 18): items[b].a at 1 prime -> items[b].dst at `target` primes, the centred residues mod q_0 re-read
// modulo every prime.  `scratch` holds [B][2][N]; h_items is the host copy of the items (the forward NTTs are launched per item).
void modraise(Context &c, u64 *scratch, const EwItem *h_items, int B, int target, hipStream_t s, const EwItem *d_items = nullptr);

// fused phase launchers (fused_ks.hip)
void f_irows_strided(const Context &c, const u64 *base, long stride, int prime_base, int period, u64 *out, long out_stride, int count,
                     hipStream_t s);
void f_irows_rot_c1(const Context &c, const KsItem *items, int ell, u64 *out, int B, hipStream_t s);
void f_irows_rs_last(const Context &c, const RsItem *items, const SumSrc *srcs, int l, u64 *out, int B, hipStream_t s);
void f_irows_rs_single(const Context &c, CtView src, int l, u64 *out, hipStream_t s);
void f_irows_decrypt(const Context &c, CtView ct, const u64 *sk, int ell, u64 *out, hipStream_t s);
// first inverse phase of a1*b1 of every item (the tensor product's c2, computed in the loader)
void f_irows_tensor_c2(const Context &c, const MulItem *items, int ell, u64 *out, int B, hipStream_t s);
// batched opcode 10: inverse ROWS phase of c0 + c1*s of every item -> out[B][ell][N] ...
void f_irows_decrypt_items(const Context &c, const BootItem *items, const SumSrc *srcs, const u64 *sk, int ell, u64 *out, int B,
                           hipStream_t s);
// re-encode + reduce + first forward phase in one launch (ell == 1: every target limb recomputes the trivial composition):
// pt [B][ell][N] coefficient domain -> ptx [B][t][N] after the COLS phase
void f_boot_reencode_fcols(const Context &c, const u64 *pt, u64 *ptx, const BootItem *items, int B, int ell, int t, CrtDev crt,
                           hipStream_t s);
// ... and the last forward phase of the re-encoded plaintexts ptx[B][t][N], added to the items' zero-encryptions
void f_frows_boot_final(const Context &c, const u64 *ptx, const BootItem *items, int B, int t, hipStream_t s);
void f_ks_icols_lift_fcols(const Context &c, const u64 *digits, u64 *ext, int B, int ell, hipStream_t s);
void f_dr_icols_lift_fcols(const Context &c, const u64 *last, long last_stride, u64 *tmp, int polys, int cnt, int l, hipStream_t s);
// L3 + L4 + L5 fused (small batches): second NTT phase of the lifted digits, inner products with the key, first inverse phase of
// the special-prime accumulators.  mode 0 rotation (items[b].key, operand through the Galois permutation) / 1 relinearisation
// fold_base (rotations): the accumulators of the data primes leave with P galois(c0) added, and f_frows_final is told so (base_folded)
void f_ks_frows_mac(const Context &c, int mode, const u64 *ext, const u64 *target, const KsItem *items, const u64 *shared_key, u64 *acc,
                    int B, int ell, hipStream_t s, bool fold_base = false);
// large-batch variants: inverse COLS phase run separately (once per source limb), then base change + forward COLS
void f_ks_lift_fcols(const Context &c, const u64 *digits, u64 *ext, int B, int ell, hipStream_t s);
void f_dr_lift_fcols(const Context &c, const u64 *last, long last_stride, u64 *tmp, int polys, int cnt, int l, hipStream_t s);
// mode 0 rotation / 1 relinearisation / 2 rescale items / 3 one rescale by value (+ optional plaintext added to c0) /
// 4 relinearisation with the c0, c1 tensor terms computed in the epilogue (small batches: no tensor launch)
void f_frows_final(const Context &c, int mode, const u64 *tmp, const void *items, const u64 *acc, int polys, int cnt, int l,
                   hipStream_t s, RsItem single = RsItem{}, const u64 *plain = nullptr, const SumSrc *srcs = nullptr,
                   const Handoff &h = Handoff{}, bool base_folded = false);
// a single rescale_to_next of `src` (level ell) into dst, optionally adding a level-(ell-1) plaintext to c0: 3 launches
void rescale_fused(Context &c, const Workspace &w, CtView dst, CtView src, int ell, const u64 *plain, hipStream_t s);

} // namespace dacapo
