// Kernel-level C ABI (include/dacapo_ckks.h): thin extern "C" shims over the launchers.
#include "../../include/dacapo_ckks.h"

#include "kernels.hpp"
#include "plan.hpp"

using namespace dacapo;

#include "c_api_types.hpp"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }
static inline CtView V(const uint64_t *p, long stride) { return CtView{ const_cast<u64 *>(p), stride }; }

// The composite ops run the same fused launch sequences as the HEVM execution plan (batch_ops.hip, fused_ks.hip) with a batch
// of one item; the item descriptor goes through a small ring of device slots, written in stream order.  Operands that alias
// the destination in a way the fused epilogues cannot tolerate fall back to the alias-safe unfused composites (ckks_ops.hip).
namespace {
constexpr int kItemSlots = 64;
constexpr size_t kItemBytes = 128;
static_assert(sizeof(KsItem) <= kItemBytes && sizeof(MulItem) <= kItemBytes && sizeof(RsItem) <= kItemBytes, "item slot too small");
void *item_slot(dc_context *ctx, const void *host_item, size_t bytes, hipStream_t s)
{
    if (!ctx->item_ring) DC_HIP_CHECK(hipMalloc(&ctx->item_ring, kItemSlots * kItemBytes));
    char *slot = static_cast<char *>(ctx->item_ring) + (size_t)(ctx->item_next++ % kItemSlots) * kItemBytes;
    DC_HIP_CHECK(hipMemcpyAsync(slot, host_item, bytes, hipMemcpyHostToDevice, s));
    return slot;
}
BatchWs batch_ws(const Workspace &w) { return BatchWs{ w.ct_tmp, w.ks_digits, w.ks_ext, w.ks_acc, w.ks_tmp }; }
bool overlaps(const uint64_t *a, const uint64_t *b) { return a == b; }
} // namespace

extern "C" {

dc_context *dc_context_create(int logN, int num_primes, int bit_size, const uint64_t *primes)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "[dacapo_amd] no HIP device: the HEVM runtime has no CPU fallback\n");
        abort();
    }
    if (!primes && (bit_size < kMinQBits || bit_size > kQBits)) {
        fprintf(stderr, "[dacapo_amd] prime width %d outside %d..%d bits (the reference's chain is 60-bit, SEAL_HEVM.cpp:48-53)\n", bit_size, kMinQBits, kQBits);
        abort();
    }
    Context *c = new Context(logN, num_primes, bit_size, primes);
    c->ensure_scratch();
    return new dc_context{ c, true };
}
// EXTENSION: a context whose key switching uses grouped digits (hybrid_ks.hip): the last `special` primes are special, a digit is `alpha`
// data primes, keys passed to dc_ct_rotate_hop / dc_ct_mul_relin / dc_keyswitch are [ceil((K - special) / alpha)][2][K][N]
dc_context *dc_context_create_hybrid(int logN, int num_primes, int special, int alpha)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "[dacapo_amd] no HIP device: the HEVM runtime has no CPU fallback\n");
        abort();
    }
    Context *c = new Context(logN, num_primes, kQBits, nullptr, special, alpha);
    c->ensure_scratch();
    return new dc_context{ c, true };
}
// ... on an explicit chain (a HEaaN-style mixed one: 60-bit base and special primes around 51-bit rescale primes; the generic-width build)
dc_context *dc_context_create_hybrid_primes(int logN, const uint64_t *primes, int num_primes, int special, int alpha)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "[dacapo_amd] no HIP device: the HEVM runtime has no CPU fallback\n");
        abort();
    }
    Context *c = new Context(logN, num_primes, kQBits, primes, special, alpha);
    c->ensure_scratch();
    return new dc_context{ c, true };
}
int dc_context_key_digits(const dc_context *ctx) { return ctx->c->key_digits(); }
int dc_context_max_level(const dc_context *ctx) { return ctx->c->max_level(); }
void dc_context_destroy(dc_context *ctx)
{
    if (ctx && ctx->item_ring) (void)hipFree(ctx->item_ring);
    if (ctx && ctx->owned) delete ctx->c;
    delete ctx;
}
int dc_context_logn(const dc_context *ctx) { return ctx->c->logN; }
int dc_context_num_primes(const dc_context *ctx) { return ctx->c->K; }
void dc_context_primes(const dc_context *ctx, uint64_t *out)
{
    for (int i = 0; i < ctx->c->K; i++) out[i] = ctx->c->primes[i];
}
void dc_context_roots(const dc_context *ctx, uint64_t *out)
{
    for (int i = 0; i < ctx->c->K; i++) out[i] = ctx->c->psi[i];
}

void *dc_malloc(size_t bytes)
{
    void *p = nullptr;
    DC_HIP_CHECK(hipMalloc(&p, bytes));
    return p;
}
void dc_free(void *dptr) { DC_HIP_CHECK(hipFree(dptr)); }
void dc_memcpy_h2d(void *dst, const void *src, size_t bytes) { DC_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); }
void dc_memcpy_d2h(void *dst, const void *src, size_t bytes) { DC_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); }
void dc_memset(void *dst, int value, size_t bytes) { DC_HIP_CHECK(hipMemset(dst, value, bytes)); }
void dc_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    DC_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
}
void dc_stream_sync(void *stream) { DC_HIP_CHECK(hipStreamSynchronize(S(stream))); }

void dc_set_device(int device) { DC_HIP_CHECK(hipSetDevice(device)); }
void dc_device_sync(void) { DC_HIP_CHECK(hipDeviceSynchronize()); }
void dc_mem_info(uint64_t *free_bytes, uint64_t *total_bytes)
{
    size_t f = 0, t = 0;
    DC_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
}
int dc_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

void *dc_event_create(void)
{
    hipEvent_t e;
    DC_HIP_CHECK(hipEventCreate(&e));
    return (void *)e;
}
void dc_event_destroy(void *event) { DC_HIP_CHECK(hipEventDestroy((hipEvent_t)event)); }
void dc_event_record(void *event, void *stream) { DC_HIP_CHECK(hipEventRecord((hipEvent_t)event, S(stream))); }
float dc_event_elapsed_ms(void *start, void *stop)
{
    float ms = 0.f;
    DC_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
    DC_HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
    return ms;
}

void dc_ntt_forward(dc_context *ctx, uint64_t *data, long limb_stride, int count, const int32_t *d_prime_idx, int prime_base,
                    int prime_period, void *stream)
{
    launch_ntt(*ctx->c, false, data, limb_stride, count, d_prime_idx, prime_base, prime_period, S(stream));
}
void dc_ntt_inverse(dc_context *ctx, uint64_t *data, long limb_stride, int count, const int32_t *d_prime_idx, int prime_base,
                    int prime_period, void *stream)
{
    launch_ntt(*ctx->c, true, data, limb_stride, count, d_prime_idx, prime_base, prime_period, S(stream));
}

// variant 0: the two-launch tiles whatever the batch size; 1: the single-crossing kernel (N = 2^15 only; aborts otherwise)
void dc_ntt_variant(dc_context *ctx, int variant, int inverse, uint64_t *data, long limb_stride, int count, const int32_t *d_prime_idx,
                    int prime_base, int prime_period, void *stream)
{
    if (variant == 1) {
        if (!ntt_full_supported(*ctx->c)) {
            fprintf(stderr, "[dacapo_amd] dc_ntt_variant: the single-crossing transform exists for N = 2^15 only\n");
            abort();
        }
        launch_ntt_full(*ctx->c, inverse != 0, data, limb_stride, count, d_prime_idx, prime_base, prime_period, S(stream));
    } else
        launch_ntt_two_phase(*ctx->c, inverse != 0, data, limb_stride, count, d_prime_idx, prime_base, prime_period, S(stream));
}

void dc_ct_negate(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, int ell, void *stream)
{
    launch_ew(*ctx->c, EwOp::Neg, V(dst, dst_stride), V(a, a_stride), V(a, a_stride), 2, 2, ell, S(stream));
}
void dc_ct_add(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, const uint64_t *b,
               long b_stride, int ell, void *stream)
{
    launch_ew(*ctx->c, EwOp::Add, V(dst, dst_stride), V(a, a_stride), V(b, b_stride), 2, 2, ell, S(stream));
}
void dc_ct_add_plain(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, const uint64_t *plain,
                     int ell, void *stream)
{
    launch_add_plain(*ctx->c, V(dst, dst_stride), V(a, a_stride), plain, ell, S(stream));
}
void dc_ct_mul_plain(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, const uint64_t *plain,
                     int ell, void *stream)
{
    launch_ew(*ctx->c, EwOp::Mul, V(dst, dst_stride), V(a, a_stride), V(plain, 0), 2, 1, ell, S(stream));
}
void dc_ct_mul_relin(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, const uint64_t *b,
                     long b_stride, const uint64_t *relin_key, int ell, void *stream)
{
    if (overlaps(dst, a) || overlaps(dst, b)) {
        mul_relin(*ctx->c, ctx->c->ws0, V(dst, dst_stride), V(a, a_stride), V(b, b_stride), relin_key, ell, S(stream));
        return;
    }
    const MulItem it{ V(a, a_stride), V(b, b_stride), V(dst, dst_stride) };
    const MulItem *d = static_cast<const MulItem *>(item_slot(ctx, &it, sizeof(it), S(stream)));
    b_mul_relin(*ctx->c, batch_ws(ctx->c->ws0), d, relin_key, 1, ell, S(stream));
}
void dc_ct_rotate_hop(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *src, long src_stride,
                      uint32_t galois_elt, const uint64_t *galois_key, int ell, void *stream)
{
    if (overlaps(dst, src)) {
        rotate_hop(*ctx->c, ctx->c->ws0, V(dst, dst_stride), V(src, src_stride), galois_elt, galois_key, ell, S(stream));
        return;
    }
    const KsItem it{ V(src, src_stride), V(dst, dst_stride), galois_key, galois_elt, 0 };
    const KsItem *d = static_cast<const KsItem *>(item_slot(ctx, &it, sizeof(it), S(stream)));
    b_rotate_hops(*ctx->c, batch_ws(ctx->c->ws0), d, 1, ell, S(stream));
}
void dc_ct_rescale(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *src, long src_stride, int ell, void *stream)
{
    const RsItem it{ V(src, src_stride), V(dst, dst_stride), 0, 0, nullptr, nullptr }; // element-wise epilogue: in place is fine
    const RsItem *d = static_cast<const RsItem *>(item_slot(ctx, &it, sizeof(it), S(stream)));
    b_rescale(*ctx->c, batch_ws(ctx->c->ws0), d, 1, ell, S(stream));
}
void dc_ct_modswitch(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *src, long src_stride, int ell, int down,
                     void *stream)
{
    if (down <= 0 || (dst == src && dst_stride == src_stride)) return; // dropping limbs in place moves nothing
    launch_ew(*ctx->c, EwOp::Copy, V(dst, dst_stride), V(src, src_stride), V(src, src_stride), 2, 2, ell - down, S(stream));
}

void dc_keyswitch(dc_context *ctx, uint64_t *out, long out_stride, const uint64_t *base0, const uint64_t *base1,
                  const uint64_t *target, const uint64_t *key, int ell, void *stream)
{
    keyswitch(*ctx->c, ctx->c->ws0, V(out, out_stride), base0, base1, target, key, ell, S(stream));
}
void dc_galois_ntt(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *src, long src_stride, uint32_t galois_elt,
                   int polys, int ell, void *stream)
{
    launch_galois(*ctx->c, V(dst, dst_stride), V(src, src_stride), galois_elt, polys, ell, S(stream));
}
void dc_poly_mul(dc_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, int ell, void *stream)
{
    launch_ew(*ctx->c, EwOp::Mul, V(dst, 0), V(a, 0), V(b, 0), 1, 1, ell, S(stream));
}
void dc_poly_add(dc_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, int ell, void *stream)
{
    launch_ew(*ctx->c, EwOp::Add, V(dst, 0), V(a, 0), V(b, 0), 1, 1, ell, S(stream));
}

uint32_t dc_galois_elt_from_step(const dc_context *ctx, int step)
{
    const uint32_t n = (uint32_t)ctx->c->N, m = 2 * n;
    if (step == 0) return m - 1;
    const uint32_t pos = (uint32_t)(step < 0 ? -step : step);
    if (pos >= (n >> 1)) return 0;
    uint32_t e = step < 0 ? (n >> 1) - pos : pos;
    uint64_t elt = 1, g = 3; // 3^e mod 2N by square and multiply
    for (; e; e >>= 1) {
        if (e & 1) elt = (elt * g) & (m - 1);
        g = (g * g) & (m - 1);
    }
    return (uint32_t)elt;
}

} // extern "C"
