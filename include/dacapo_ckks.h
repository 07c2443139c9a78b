/*
 * dacapo_ckks.h -- kernel-level C ABI of the MI355X HEVM runtime (libSEAL_HEVM.so, same library that exports
 * the 18 HEVM symbols of include/hevm_abi.h).
 *
 * Each entry point stands for one call that /root/reference/lib/Runtime/SEAL_HEVM.cpp makes into Microsoft
 * SEAL 4.0.0 (the reference's FFI boundary for the arithmetic of the hot path); the citation after each
 * declaration is the reference line that makes the call.  Plain pointers and sizes only: polynomial data are
 * DEVICE pointers to uint64 limbs, `stream` is a hipStream_t (NULL = default stream).  Nothing here
 * synchronises or allocates unless it says so.  A missing GPU / failed HIP call aborts the process with a
 * message, like the reference's asserts and uncaught SEAL exceptions do (SEAL_HEVM.cpp:295,327,496).
 *
 * Layouts (limb-major, N coefficients per limb, NTT domain = SEAL's bit-reversed evaluation order):
 *   polynomial at level ell : [ell][N]                 limb i is modulo prime i of the chain
 *   ciphertext              : 2 polynomials, `poly_stride` elements apart (poly_stride >= ell*N)
 *   key-switch key          : [K-1 digits][2][K][N]    K = primes in the key-level chain, special prime = K-1
 */
#ifndef DACAPO_CKKS_H
#define DACAPO_CKKS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: what this header declares is what it exports */
#endif

typedef struct dc_context dc_context;

/* seal::EncryptionParameters + SEALContext: N = 2^logN, `num_primes` primes.  primes == NULL builds
 * CoeffModulus::Create(N, {bit_size x num_primes}) exactly as SEAL_HEVM.cpp:48-53 does (bit_size must be 60:
 * the HIP reduction is specialised to q = 2^60 - delta, delta < 2^28).  Uploads twiddle tables.  Synchronous. */
dc_context *dc_context_create(int logN, int num_primes, int bit_size, const uint64_t *primes);
/* EXTENSION (not SEAL's scheme; the reference's HEaaN runtime has it inside its closed library, HEAAN_HEVM.cpp:124-141): key switching
 * with grouped digits -- the last `special` primes of the chain are special, a decomposition digit is a group of `alpha` data primes
 * (alpha <= special), key-switch keys are [dc_context_key_digits()][2][K][N] and data levels run up to dc_context_max_level() =
 * num_primes - special.  special = alpha = 1 is dc_context_create's (SEAL's) scheme. */
dc_context *dc_context_create_hybrid(int logN, int num_primes, int special, int alpha);
/* ... on an explicit prime chain (each prime 2^b - d as for dc_context_create; widths other than 60 bits: libSEAL_HEVM_gw.so) */
dc_context *dc_context_create_hybrid_primes(int logN, const uint64_t *primes, int num_primes, int special, int alpha);
int dc_context_key_digits(const dc_context *ctx);
int dc_context_max_level(const dc_context *ctx);
void dc_context_destroy(dc_context *ctx);
int dc_context_logn(const dc_context *ctx);
int dc_context_num_primes(const dc_context *ctx);
void dc_context_primes(const dc_context *ctx, uint64_t *out /* [num_primes] host */);
void dc_context_roots(const dc_context *ctx, uint64_t *out /* [num_primes] host: minimal primitive 2N-th roots */);

/* device memory plumbing so that callers need no HIP/torch types (all synchronous) */
void *dc_malloc(size_t bytes);
void dc_free(void *dptr);
void dc_memcpy_h2d(void *dst, const void *src, size_t bytes);
void dc_memcpy_d2h(void *dst, const void *src, size_t bytes);
void dc_memset(void *dst, int value, size_t bytes);
void dc_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream); /* asynchronous on `stream` */
void dc_stream_sync(void *stream);
void dc_set_device(int device);  /* hipSetDevice: one process per GPU, call before creating contexts/VMs */
void dc_device_sync(void);       /* hipDeviceSynchronize */
int dc_device_count(void);
/* free and total HBM of the current device in bytes (hipMemGetInfo) */
void dc_mem_info(uint64_t *free_bytes, uint64_t *total_bytes);
/* HIP events on `stream`, for timing the kernels where they are launched (bench.py) */
void *dc_event_create(void);
void dc_event_destroy(void *event);
void dc_event_record(void *event, void *stream);
float dc_event_elapsed_ms(void *start, void *stop); /* synchronises on `stop` */

/* util::ntt_negacyclic_harvey / inverse_ntt_negacyclic_harvey over `count` limbs at data + b*limb_stride, in place.
 * Limb b is modulo prime d_prime_idx[b] (DEVICE int32 array) or, if NULL, prime_base + (b % prime_period)
 * (prime_period <= 0: prime_base + b).  Reached from every rotate/rescale/mulcc/encode of SEAL_HEVM.cpp
 * (:262,:273,:283,:316). */
void dc_ntt_forward(dc_context *ctx, uint64_t *data, long limb_stride, int count, const int32_t *d_prime_idx, int prime_base,
                    int prime_period, void *stream);
void dc_ntt_inverse(dc_context *ctx, uint64_t *data, long limb_stride, int count, const int32_t *d_prime_idx, int prime_base,
                    int prime_period, void *stream);

/* The same transform with the implementation named (tests and bench.py compare them; dc_ntt_forward/inverse choose by batch size):
 * variant 0 = two launches per transform (ntt_tile.hpp), 1 = one 1024-thread workgroup per limb, one HBM crossing (ntt_full.hip,
 * N = 2^15 only). */
void dc_ntt_variant(dc_context *ctx, int variant, int inverse, uint64_t *data, long limb_stride, int count, const int32_t *d_prime_idx,
                    int prime_base, int prime_period, void *stream);

/* Evaluator::negate            SEAL_HEVM.cpp:278 */
void dc_ct_negate(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, int ell, void *stream);
/* Evaluator::add               SEAL_HEVM.cpp:302 */
void dc_ct_add(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, const uint64_t *b,
               long b_stride, int ell, void *stream);
/* Evaluator::add_plain         SEAL_HEVM.cpp:309 */
void dc_ct_add_plain(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, const uint64_t *plain,
                     int ell, void *stream);
/* Evaluator::multiply_plain    SEAL_HEVM.cpp:322 */
void dc_ct_mul_plain(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, const uint64_t *plain,
                     int ell, void *stream);
/* Evaluator::multiply + relinearize_inplace   SEAL_HEVM.cpp:315-316 */
void dc_ct_mul_relin(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *a, long a_stride, const uint64_t *b,
                     long b_stride, const uint64_t *relin_key, int ell, void *stream);
/* Evaluator::apply_galois_inplace: one hop of Evaluator::rotate_vector   SEAL_HEVM.cpp:273 */
void dc_ct_rotate_hop(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *src, long src_stride,
                      uint32_t galois_elt, const uint64_t *galois_key, int ell, void *stream);
/* Evaluator::rescale_to_next: level ell -> ell-1   SEAL_HEVM.cpp:283 */
void dc_ct_rescale(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *src, long src_stride, int ell,
                   void *stream);
/* Evaluator::mod_switch_to_next (CKKS: drop the last `down` limbs = copy the kept ones)   SEAL_HEVM.cpp:289-291 */
void dc_ct_modswitch(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *src, long src_stride, int ell, int down,
                     void *stream);

/* building blocks exposed for the parity tests */
/* Evaluator::switch_key_inplace: (out0,out1) = (base0,base1) + KS(target); base pointers may be NULL (= 0). */
void dc_keyswitch(dc_context *ctx, uint64_t *out, long out_stride, const uint64_t *base0, const uint64_t *base1,
                  const uint64_t *target, const uint64_t *key, int ell, void *stream);
/* GaloisTool::apply_galois_ntt on `polys` polynomials of `ell` limbs */
void dc_galois_ntt(dc_context *ctx, uint64_t *dst, long dst_stride, const uint64_t *src, long src_stride, uint32_t galois_elt,
                   int polys, int ell, void *stream);
/* limb-wise dyadic product / sum of two [ell][N] polynomials (dyadic_product_coeffmod / add_poly_coeffmod) */
void dc_poly_mul(dc_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, int ell, void *stream);
void dc_poly_add(dc_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, int ell, void *stream);

/* GaloisTool::get_elt_from_step (host): step > 0 rotates left; 0 = conjugation; returns 0 if |step| >= N/2 */
uint32_t dc_galois_elt_from_step(const dc_context *ctx, int step);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
