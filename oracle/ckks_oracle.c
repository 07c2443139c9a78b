/*
 * ckks_oracle.c -- CPU ORACLE for the HEVM / SEAL RNS-CKKS hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (dacapo_amd/lib/libSEAL_HEVM.so) never
 * links, loads or calls anything in this directory.
 *
 * What it restates: the arithmetic that /root/reference/lib/Runtime/SEAL_HEVM.cpp reaches through
 * Microsoft SEAL 4.0.0 (README.md:65-73, versions.txt:4, CMakeLists.txt:60).  SEAL's sources are NOT in
 * /root/reference (external dependency, not vendored) and SEAL is not installed in this image, so the
 * algorithms below are restated from SEAL 4.0.0's published algorithm (marked [SEAL-upstream]) and each
 * function cites the reference call site that reaches it.
 *
 * PARITY STATUS: "parity unpinned" against SEAL at the limb level -- the reference repository ships
 * no golden vectors, no unit tests and no .hevm/.cst artefacts for this path.
 * What pins this oracle instead (tests/test_oracle_*.py):
 *   - the prime chain of CoeffModulus::Create(2^15, {60 x 14}) begins/ends with SEAL's well known
 *     60-bit primes (0xffffffffffc0001 is SEAL's first 60-bit NTT prime for 2N = 2^16);
 *   - known-answer values of SEAL's own unit test tests/seal/util/ntt.cpp (NTTTablesTest: minimal
 *     primitive 2N-th roots mod 0xffffffffffc0001 in bit-reversed table order), recalled from
 *     SEAL upstream and re-verified here algebraically (they are primitive roots, minimal, and the
 *     table order matches);
 *   - forward NTT == O(N^2) negacyclic evaluation, inverse(forward(x)) == x, dyadic product ==
 *     schoolbook negacyclic convolution;
 *   - two independent implementations (plain `%` on unsigned __int128 vs Harvey/Shoup/Barrett lazy
 *     arithmetic) agree bit for bit;
 *   - closed forms on canonical representatives for rescale / key-switch mod-down;
 *   - decrypt(op(encrypt x)) ~= op(x) for every HEVM opcode.
 *
 * All outputs are canonical residues in [0, q), so any correct implementation (SEAL's included) must
 * produce identical limbs for identical limb inputs, whatever its lazy-reduction strategy.
 *
 * Layouts (all uint64, little endian, NTT domain = SEAL's bit-reversed evaluation order):
 *   polynomial at level ell : [ell][N]           (limb i is modulo prime index i)
 *   ciphertext              : [2][ell][N]
 *   key-switch key          : [K-1 digits][2][K][N]   (K = number of key primes, special prime = K-1)
 */
#include <complex.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#undef I /* complex.h's imaginary unit macro; _Complex_I is used instead */

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define ORC_MAX_PRIMES 64

typedef struct {
    u64 q;
    u64 ratio_lo, ratio_hi; /* floor(2^128 / q), SEAL Modulus::const_ratio [SEAL-upstream] */
} orc_mod;

typedef struct orc_ctx {
    int logN;
    size_t N;
    int K; /* number of primes in the key-level chain (data primes 0..K-2, special prime K-1) */
    int ks, alpha; /* grouped-digit extension (orc_set_hybrid): ks special primes K-ks..K-1, digits of alpha data primes; 1, 1 = SEAL */
    orc_mod mod[ORC_MAX_PRIMES];
    u64 psi[ORC_MAX_PRIMES];     /* minimal primitive 2N-th root */
    u64 *rp[ORC_MAX_PRIMES];     /* rp[k]  = psi^{bitrev(k)}           (SEAL NTTTables::root_powers_) */
    u64 *rp_sh[ORC_MAX_PRIMES];  /* floor(rp * 2^64 / q)               (MultiplyUIntModOperand.quotient) */
    u64 *irp[ORC_MAX_PRIMES];    /* irp[k] = rp[k]^{-1}; see note in orc_ntt_inv about SEAL's layout */
    u64 *irp_sh[ORC_MAX_PRIMES];
    u64 inv_n[ORC_MAX_PRIMES], inv_n_sh[ORC_MAX_PRIMES];
    double complex *croot; /* croot[k] = exp(2*pi*i*bitrev(k)/(2N))    (CKKSEncoder::root_powers_) */
    uint32_t *slot_map;    /* CKKSEncoder::matrix_reps_index_map_ */
} orc_ctx;

/* SEAL's evaluator and the reference's HEVM loop are single-threaded, and so is this file by default.  For the
 * "8-thread OpenMP over limbs" comparison of BASELINE.md the limb-parallel loops of key switching, rescale and the batched transforms
 * carry `omp parallel for` with `if (orc_threads > 1)`: same arithmetic per limb, hence the same result bits at any thread count. */
static int orc_threads = 1;
void orc_set_threads(int n)
{
    orc_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
    omp_set_num_threads(orc_threads);
#endif
}
int orc_get_threads(void) { return orc_threads; }
int orc_has_openmp(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * Scalar modular arithmetic
 * ---------------------------------------------------------------------------------------------- */
static inline u64 mulmod_simple(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }

static u64 powmod(u64 a, u64 e, u64 q)
{
    u64 r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = mulmod_simple(r, a, q);
        a = mulmod_simple(a, a, q);
        e >>= 1;
    }
    return r;
}

static u64 invmod_prime(u64 a, u64 q) { return powmod(a, q - 2, q); }

/* Barrett reduction of a 128-bit value with the two-word ratio, as SEAL util::barrett_reduce_128
 * [SEAL-upstream uintarithsmallmod.h]; result in [0,q). */
static inline u64 barrett128(u128 x, const orc_mod *m)
{
    u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    u64 c = (u64)(((u128)x0 * m->ratio_lo) >> 64);
    u128 mid = (u128)x0 * m->ratio_hi + c;
    u128 mid2 = (u128)x1 * m->ratio_lo + (u64)mid;
    u64 qhat = x1 * m->ratio_hi + (u64)(mid >> 64) + (u64)(mid2 >> 64);
    u64 r = x0 - qhat * m->q;
    return r - (m->q & (u64)(-(int64_t)(r >= m->q)));
}

/* Barrett reduction of a 64-bit value (SEAL util::barrett_reduce_64): result in [0,q). */
static inline u64 barrett64(u64 x, const orc_mod *m)
{
    u64 qhat = (u64)(((u128)x * m->ratio_hi) >> 64);
    u64 r = x - qhat * m->q;
    return r - (m->q & (u64)(-(int64_t)(r >= m->q)));
}

static inline u64 mulmod(u64 a, u64 b, const orc_mod *m) { return barrett128((u128)a * b, m); }
static inline u64 addmod(u64 a, u64 b, u64 q)
{
    u64 s = a + b;
    return s - (q & (u64)(-(int64_t)(s >= q)));
}
static inline u64 submod(u64 a, u64 b, u64 q)
{
    u64 d = a - b;
    return d + (q & (u64)(-(int64_t)(a < b)));
}
static inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

static inline u64 shoup_of(u64 w, u64 q) { return (u64)((((u128)w) << 64) / q); }
/* w*y mod q, lazy: result in [0, 2q) for any 64-bit y (SEAL multiply_uint_mod_lazy) */
static inline u64 mul_shoup_lazy(u64 y, u64 w, u64 wsh, u64 q)
{
    u64 Q = (u64)(((u128)wsh * y) >> 64);
    return w * y - Q * q;
}

/* ------------------------------------------------------------------------------------------------
 * Primes and roots of unity  [SEAL-upstream numth.cpp: is_prime / get_primes /
 * try_minimal_primitive_root; modulus.cpp: CoeffModulus::Create] -- reached from
 * SEAL_HEVM.cpp:48-53 (parms.set_coeff_modulus(CoeffModulus::Create(1<<15, {60 x 14}))).
 * ---------------------------------------------------------------------------------------------- */
int orc_is_prime(u64 n)
{
    static const u64 bases[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
    if (n < 2) return 0;
    for (size_t i = 0; i < sizeof(bases) / sizeof(bases[0]); i++) {
        if (n == bases[i]) return 1;
        if (n % bases[i] == 0) return 0;
    }
    u64 d = n - 1;
    int r = 0;
    while (!(d & 1)) {
        d >>= 1;
        r++;
    }
    for (size_t i = 0; i < sizeof(bases) / sizeof(bases[0]); i++) { /* deterministic for n < 2^64 */
        u64 x = powmod(bases[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int j = 1; j < r; j++) {
            x = mulmod_simple(x, x, n);
            if (x == n - 1) {
                comp = 0;
                break;
            }
        }
        if (comp) return 0;
    }
    return 1;
}

/* SEAL util::get_primes(factor, bit_size, count): scan DOWN from the largest value == 1 (mod factor)
 * below 2^bit_size; returned in the order found (largest first). Returns number found. */
int orc_get_primes(u64 factor, int bit_size, int count, u64 *out)
{
    u64 value = ((((u64)1) << bit_size) - 1) / factor * factor + 1;
    u64 lower = ((u64)1) << (bit_size - 1);
    int found = 0;
    while (found < count && value > lower) {
        if (orc_is_prime(value)) out[found++] = value;
        value -= factor;
    }
    return found;
}

/* CoeffModulus::Create(N, {bits x count}) for equal bit sizes: result[i] = found[count-1-i]
 * (each size's primes are popped from the back), so q_0 is the LAST prime found and the special
 * prime (last entry) is the FIRST found. */
int orc_coeff_modulus_create(int logN, int bit_size, int count, u64 *out)
{
    u64 tmp[ORC_MAX_PRIMES];
    if (count > ORC_MAX_PRIMES) return -1;
    int found = orc_get_primes(((u64)2) << logN, bit_size, count, tmp);
    if (found != count) return -1;
    for (int i = 0; i < count; i++) out[i] = tmp[count - 1 - i];
    return 0;
}

/* smallest primitive degree-th root of unity mod q (degree = power of two) */
u64 orc_min_primitive_root(u64 degree, u64 q)
{
    if ((q - 1) % degree) return 0;
    u64 e = (q - 1) / degree, g = 0;
    for (u64 x = 2; x < 1000; x++) {
        g = powmod(x, e, q);
        if (powmod(g, degree / 2, q) == q - 1) break; /* primitive iff g^(degree/2) == -1 */
        g = 0;
    }
    if (!g) return 0;
    u64 gsq = mulmod_simple(g, g, q), cur = g, best = g;
    for (u64 i = 0; i < degree / 2; i++) { /* all odd powers = all primitive roots */
        if (cur < best) best = cur;
        cur = mulmod_simple(cur, gsq, q);
    }
    return best;
}

static inline uint32_t bitrev32(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
uint32_t orc_bitrev(uint32_t x, int bits) { return bitrev32(x, bits); }

/* ------------------------------------------------------------------------------------------------
 * Context
 * ---------------------------------------------------------------------------------------------- */
void orc_destroy(orc_ctx *c)
{
    if (!c) return;
    for (int i = 0; i < c->K; i++) {
        free(c->rp[i]);
        free(c->rp_sh[i]);
        free(c->irp[i]);
        free(c->irp_sh[i]);
    }
    free(c->croot);
    free(c->slot_map);
    free(c);
}

/* primes == NULL: build the SEAL chain CoeffModulus::Create(2^logN, {bit_size x K}). */
orc_ctx *orc_create(int logN, int K, int bit_size, const u64 *primes)
{
    if (K < 1 || K > ORC_MAX_PRIMES || logN < 1 || logN > 20) return NULL;
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    c->logN = logN;
    c->N = ((size_t)1) << logN;
    c->K = K;
    c->ks = c->alpha = 1;
    u64 chain[ORC_MAX_PRIMES];
    if (primes)
        memcpy(chain, primes, sizeof(u64) * K);
    else if (orc_coeff_modulus_create(logN, bit_size, K, chain)) {
        free(c);
        return NULL;
    }
    size_t N = c->N;
    for (int i = 0; i < K; i++) {
        u64 q = chain[i];
        c->mod[i].q = q;
        u128 ratio = (~(u128)0) / q; /* == floor(2^128/q) for odd q > 1 */
        c->mod[i].ratio_lo = (u64)ratio;
        c->mod[i].ratio_hi = (u64)(ratio >> 64);
        u64 psi = orc_min_primitive_root(2 * (u64)N, q);
        if (!psi) {
            orc_destroy(c);
            return NULL;
        }
        c->psi[i] = psi;
        c->rp[i] = (u64 *)malloc(N * 8);
        c->rp_sh[i] = (u64 *)malloc(N * 8);
        c->irp[i] = (u64 *)malloc(N * 8);
        c->irp_sh[i] = (u64 *)malloc(N * 8);
        /* NTTTables::initialize: root_powers_[bitrev(i)] = psi^i  [SEAL-upstream ntt.cpp] */
        u64 pw = 1, ipsi = invmod_prime(psi, q), ipw = 1;
        for (size_t k = 0; k < N; k++) {
            size_t r = bitrev32((uint32_t)k, logN);
            c->rp[i][r] = pw;
            c->irp[i][r] = ipw;
            pw = mulmod_simple(pw, psi, q);
            ipw = mulmod_simple(ipw, ipsi, q);
        }
        for (size_t k = 0; k < N; k++) {
            c->rp_sh[i][k] = shoup_of(c->rp[i][k], q);
            c->irp_sh[i][k] = shoup_of(c->irp[i][k], q);
        }
        c->inv_n[i] = invmod_prime((u64)N % q, q);
        c->inv_n_sh[i] = shoup_of(c->inv_n[i], q);
    }
    /* CKKSEncoder tables [SEAL-upstream ckks.cpp ctor], reached from SEAL_HEVM.cpp:127,154,178 */
    c->croot = (double complex *)malloc(N * sizeof(double complex));
    for (size_t k = 0; k < N; k++) {
        long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)bitrev32((uint32_t)k, logN) /
                          (long double)(2 * N);
        c->croot[k] = (double)cosl(ang) + (double)sinl(ang) * _Complex_I;
    }
    c->slot_map = (uint32_t *)malloc(N * sizeof(uint32_t));
    size_t slots = N >> 1;
    u64 m = 2 * (u64)N, pos = 1;
    for (size_t i = 0; i < slots; i++) {
        uint32_t index1 = (uint32_t)((pos - 1) >> 1);
        uint32_t index2 = (uint32_t)((m - pos - 1) >> 1);
        c->slot_map[i] = bitrev32(index1, logN);
        c->slot_map[slots | i] = bitrev32(index2, logN);
        pos = (pos * 3) & (m - 1);
    }
    return c;
}

int orc_logn(const orc_ctx *c) { return c->logN; }
int orc_num_primes(const orc_ctx *c) { return c->K; }
void orc_primes(const orc_ctx *c, u64 *out)
{
    for (int i = 0; i < c->K; i++) out[i] = c->mod[i].q;
}
u64 orc_psi(const orc_ctx *c, int p) { return c->psi[p]; }
void orc_root_powers(const orc_ctx *c, int p, u64 *out) { memcpy(out, c->rp[p], c->N * 8); }
void orc_inv_root_powers(const orc_ctx *c, int p, u64 *out) { memcpy(out, c->irp[p], c->N * 8); }

/* ------------------------------------------------------------------------------------------------
 * Negacyclic NTT  [SEAL-upstream ntt.cpp / dwthandler.h: ntt_negacyclic_harvey,
 * inverse_ntt_negacyclic_harvey].  Forward: Cooley-Tukey, natural-order input, bit-reversed output,
 * stage with m groups uses root_powers[m + i] for group i.  out[k] = sum_j a_j psi^{(2 bitrev(k)+1) j}.
 * ---------------------------------------------------------------------------------------------- */
void orc_ntt_fwd(const orc_ctx *c, int p, u64 *a)
{
    const u64 q = c->mod[p].q, two_q = 2 * q;
    const u64 *w = c->rp[p], *wsh = c->rp_sh[p];
    size_t n = c->N;
    for (size_t m = 1, gap = n >> 1; m < n; m <<= 1, gap >>= 1) {
        for (size_t i = 0; i < m; i++) {
            u64 W = w[m + i], Wsh = wsh[m + i];
            u64 *x = a + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                u64 u = x[j] - (two_q & (u64)(-(int64_t)(x[j] >= two_q))); /* [0,4q) -> [0,2q) */
                u64 v = mul_shoup_lazy(y[j], W, Wsh, q);                    /* [0,2q)  */
                x[j] = u + v;
                y[j] = u - v + two_q;
            }
        }
    }
    for (size_t j = 0; j < n; j++) { /* [0,4q) -> [0,q) */
        u64 v = a[j];
        v -= two_q & (u64)(-(int64_t)(v >= two_q));
        v -= q & (u64)(-(int64_t)(v >= q));
        a[j] = v;
    }
}

/* Inverse: Gentleman-Sande, bit-reversed input, natural output, then multiply by N^{-1}.
 * SEAL stores its inverse table so that it is consumed sequentially (inv_root_powers_[bitrev(i-1)+1]
 * = psi^{-i}); the factor used by the stage with m groups, group i, is exactly rp[m+i]^{-1}, which is
 * what irp[m+i] holds here -- same mathematics, different table order. */
void orc_ntt_inv(const orc_ctx *c, int p, u64 *a)
{
    const u64 q = c->mod[p].q, two_q = 2 * q;
    const u64 *w = c->irp[p], *wsh = c->irp_sh[p];
    size_t n = c->N;
    for (size_t m = n >> 1, gap = 1; m >= 1; m >>= 1, gap <<= 1) {
        for (size_t i = 0; i < m; i++) {
            u64 W = w[m + i], Wsh = wsh[m + i];
            u64 *x = a + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                u64 u = x[j], v = y[j]; /* both in [0,2q) */
                u64 s = u + v;
                x[j] = s - (two_q & (u64)(-(int64_t)(s >= two_q)));
                y[j] = mul_shoup_lazy(u - v + two_q, W, Wsh, q);
            }
        }
    }
    u64 inv = c->inv_n[p], invsh = c->inv_n_sh[p];
    for (size_t j = 0; j < n; j++) {
        u64 v = mul_shoup_lazy(a[j], inv, invsh, q);
        a[j] = v - (q & (u64)(-(int64_t)(v >= q)));
    }
}

/* Independent second implementation: plain % arithmetic, no lazy ranges (cross-check only). */
void orc_ntt_fwd_simple(const orc_ctx *c, int p, u64 *a)
{
    const u64 q = c->mod[p].q;
    size_t n = c->N;
    for (size_t m = 1, gap = n >> 1; m < n; m <<= 1, gap >>= 1)
        for (size_t i = 0; i < m; i++) {
            u64 W = c->rp[p][m + i];
            for (size_t j = 2 * i * gap; j < 2 * i * gap + gap; j++) {
                u64 u = a[j], v = mulmod_simple(a[j + gap], W, q);
                a[j] = (u + v) % q;
                a[j + gap] = (u + q - v) % q;
            }
        }
}
void orc_ntt_inv_simple(const orc_ctx *c, int p, u64 *a)
{
    const u64 q = c->mod[p].q;
    size_t n = c->N;
    for (size_t m = n >> 1, gap = 1; m >= 1; m >>= 1, gap <<= 1)
        for (size_t i = 0; i < m; i++) {
            u64 W = c->irp[p][m + i];
            for (size_t j = 2 * i * gap; j < 2 * i * gap + gap; j++) {
                u64 u = a[j], v = a[j + gap];
                a[j] = (u + v) % q;
                a[j + gap] = mulmod_simple((u + q - v) % q, W, q);
            }
        }
    for (size_t j = 0; j < n; j++) a[j] = mulmod_simple(a[j], c->inv_n[p], q);
}

/* O(N^2) definition: out[k] = sum_j a_j * psi^{(2*bitrev(k)+1)*j} mod q.  Small N only. */
void orc_ntt_fwd_definition(const orc_ctx *c, int p, const u64 *a, u64 *out)
{
    const u64 q = c->mod[p].q;
    size_t n = c->N;
    for (size_t k = 0; k < n; k++) {
        u64 e = 2 * (u64)bitrev32((uint32_t)k, c->logN) + 1;
        u64 base = powmod(c->psi[p], e, q), pw = 1, acc = 0;
        for (size_t j = 0; j < n; j++) {
            acc = (acc + mulmod_simple(a[j] % q, pw, q)) % q;
            pw = mulmod_simple(pw, base, q);
        }
        out[k] = acc;
    }
}

/* O(N^2) negacyclic product in the coefficient domain (X^N = -1).  Small N only. */
void orc_negacyclic_schoolbook(const orc_ctx *c, int p, const u64 *a, const u64 *b, u64 *out)
{
    const u64 q = c->mod[p].q;
    size_t n = c->N;
    memset(out, 0, n * 8);
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) {
            u64 t = mulmod_simple(a[i] % q, b[j] % q, q);
            size_t k = i + j;
            if (k < n)
                out[k] = (out[k] + t) % q;
            else
                out[k - n] = (out[k - n] + q - t) % q;
        }
}

/* batched over an explicit prime-index list: limb b uses prime pidx[b] */
void orc_ntt_fwd_batch(const orc_ctx *c, const int32_t *pidx, int count, u64 *data)
{
#pragma omp parallel for if (orc_threads > 1) schedule(dynamic)
    for (int b = 0; b < count; b++) orc_ntt_fwd(c, pidx[b], data + (size_t)b * c->N);
}
void orc_ntt_inv_batch(const orc_ctx *c, const int32_t *pidx, int count, u64 *data)
{
#pragma omp parallel for if (orc_threads > 1) schedule(dynamic)
    for (int b = 0; b < count; b++) orc_ntt_inv(c, pidx[b], data + (size_t)b * c->N);
}

/* ------------------------------------------------------------------------------------------------
 * Limb-wise ("dyadic") polynomial ops at level ell (limb i <-> prime i).
 * SEAL polyarithsmallmod.h: add_poly_coeffmod / negate_poly_coeffmod / dyadic_product_coeffmod.
 * Reached from SEAL_HEVM.cpp:278 (negate), :302 (add), :309 (add_plain), :315 (multiply),
 * :322 (multiply_plain).
 * ---------------------------------------------------------------------------------------------- */
void orc_poly_add(const orc_ctx *c, int ell, const u64 *a, const u64 *b, u64 *out)
{
#pragma omp parallel for if (orc_threads > 1)
    for (int i = 0; i < ell; i++) {
        u64 q = c->mod[i].q;
        for (size_t j = 0; j < c->N; j++) out[i * c->N + j] = addmod(a[i * c->N + j], b[i * c->N + j], q);
    }
}
void orc_poly_sub(const orc_ctx *c, int ell, const u64 *a, const u64 *b, u64 *out)
{
#pragma omp parallel for if (orc_threads > 1)
    for (int i = 0; i < ell; i++) {
        u64 q = c->mod[i].q;
        for (size_t j = 0; j < c->N; j++) out[i * c->N + j] = submod(a[i * c->N + j], b[i * c->N + j], q);
    }
}
void orc_poly_neg(const orc_ctx *c, int ell, const u64 *a, u64 *out)
{
    for (int i = 0; i < ell; i++) {
        u64 q = c->mod[i].q;
        for (size_t j = 0; j < c->N; j++) out[i * c->N + j] = negmod(a[i * c->N + j], q);
    }
}
void orc_poly_mul(const orc_ctx *c, int ell, const u64 *a, const u64 *b, u64 *out)
{
#pragma omp parallel for if (orc_threads > 1)
    for (int i = 0; i < ell; i++) {
        const orc_mod *m = &c->mod[i];
        for (size_t j = 0; j < c->N; j++) out[i * c->N + j] = mulmod(a[i * c->N + j], b[i * c->N + j], m);
    }
}
/* EXTENSION (the GPU VM's option hyb_double_hoist): the same dyadic product on limbs of primes first ... first + count - 1 -- the special-prime
 * limbs of an accumulator in the raised basis times a plaintext's limbs over those primes */
void orc_poly_mul_at(const orc_ctx *c, int first, int count, const u64 *a, const u64 *b, u64 *out)
{
    for (int i = 0; i < count; i++) {
        const orc_mod *m = &c->mod[first + i];
        for (size_t j = 0; j < c->N; j++) out[i * c->N + j] = mulmod(a[i * c->N + j], b[i * c->N + j], m);
    }
}
/* same with % (cross-check) */
void orc_poly_mul_simple(const orc_ctx *c, int ell, const u64 *a, const u64 *b, u64 *out)
{
    for (int i = 0; i < ell; i++) {
        u64 q = c->mod[i].q;
        for (size_t j = 0; j < c->N; j++) out[i * c->N + j] = mulmod_simple(a[i * c->N + j], b[i * c->N + j], q);
    }
}

/* Evaluator::multiply (ckks_multiply), size-2 x size-2 -> size-3, NTT domain:
 * c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1  (SEAL_HEVM.cpp:315).  out: [3][ell][N] */
void orc_ct_tensor(const orc_ctx *c, int ell, const u64 *a, const u64 *b, u64 *out)
{
    size_t N = c->N, P = (size_t)ell * N;
    for (int i = 0; i < ell; i++) {
        const orc_mod *m = &c->mod[i];
        for (size_t j = 0; j < N; j++) {
            size_t k = i * N + j;
            u64 a0 = a[k], a1 = a[P + k], b0 = b[k], b1 = b[P + k];
            out[k] = mulmod(a0, b0, m);
            out[P + k] = addmod(mulmod(a0, b1, m), mulmod(a1, b0, m), m->q);
            out[2 * P + k] = mulmod(a1, b1, m);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Galois  [SEAL-upstream galois.cpp: GaloisTool::get_elt_from_step / generate_table_ntt /
 * apply_galois_ntt; util naf()] -- reached from SEAL_HEVM.cpp:273 (rotate_vector).
 * ---------------------------------------------------------------------------------------------- */
/* step>0 rotates left. Returns 0 on invalid step (|step| >= N/2). step==0 -> conjugation elt 2N-1. */
uint32_t orc_elt_from_step(const orc_ctx *c, int step)
{
    uint32_t n = (uint32_t)c->N, m = 2 * n;
    if (step == 0) return m - 1;
    uint32_t pos = (uint32_t)(step < 0 ? -step : step);
    if (pos >= (n >> 1)) return 0;
    uint32_t s = step < 0 ? (n >> 1) - pos : pos;
    u64 elt = 1;
    for (uint32_t i = 0; i < s; i++) elt = (elt * 3) & (m - 1);
    return (uint32_t)elt;
}

/* GaloisTool::get_elts_all(): 2N-1, then 3^(2^i), 3^-(2^i) for i < logN-1.  Returns count. */
int orc_default_galois_elts(const orc_ctx *c, uint32_t *out)
{
    u64 m = 2 * (u64)c->N;
    int cnt = 0;
    out[cnt++] = (uint32_t)(m - 1);
    u64 pos = 3, neg = 1;
    /* inverse of 3 mod 2^k */
    for (u64 x = 1; x < m; x += 2)
        if (((x * 3) & (m - 1)) == 1) {
            neg = x;
            break;
        }
    for (int i = 0; i < c->logN - 1; i++) {
        out[cnt++] = (uint32_t)pos;
        pos = (pos * pos) & (m - 1);
        out[cnt++] = (uint32_t)neg;
        neg = (neg * neg) & (m - 1);
    }
    return cnt;
}

/* util::naf(value): non-adjacent form, least significant digit first.  Returns count. */
int orc_naf(int value, int *out)
{
    int sign = value < 0, cnt = 0;
    if (sign) value = -value;
    for (int i = 0; value; i++) {
        int zi = (value & 1) ? 2 - (value & 3) : 0;
        value = (value - zi) >> 1;
        if (zi) out[cnt++] = (sign ? -zi : zi) * (1 << i);
    }
    return cnt;
}

/* generate_table_ntt: table[i] = bitrev(((elt * bitrev_{logN+1}(i + N)) >> 1) & (N-1)) */
void orc_galois_table(const orc_ctx *c, uint32_t elt, uint32_t *table)
{
    uint32_t n = (uint32_t)c->N;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t reversed = bitrev32(i + n, c->logN + 1);
        u64 index_raw = (((u64)elt * reversed) >> 1) & (n - 1);
        table[i] = bitrev32((uint32_t)index_raw, c->logN);
    }
}

/* apply_galois_ntt on `limbs` limbs: out[i] = in[table[i]] (out must not alias in) */
void orc_galois_ntt(const orc_ctx *c, uint32_t elt, int limbs, const u64 *in, u64 *out)
{
    size_t N = c->N;
    uint32_t *table = (uint32_t *)malloc(N * sizeof(uint32_t));
    orc_galois_table(c, elt, table);
    for (int l = 0; l < limbs; l++)
        for (size_t i = 0; i < N; i++) out[l * N + i] = in[l * N + table[i]];
    free(table);
}

/* coefficient-domain automorphism a(X) -> a(X^elt) mod (X^N+1, q) -- used only to cross-check the
 * NTT-domain table (SEAL apply_galois, non-NTT form). */
void orc_galois_coeff(const orc_ctx *c, int p, uint32_t elt, const u64 *in, u64 *out)
{
    size_t N = c->N;
    u64 q = c->mod[p].q, m = 2 * N;
    for (size_t i = 0; i < N; i++) {
        u64 idx = ((u64)i * elt) & (m - 1);
        u64 v = in[i];
        if (idx >= N)
            out[idx - N] = negmod(v, q);
        else
            out[idx] = v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * divide-and-round by the last prime of a basis, NTT domain
 * [SEAL-upstream rns.cpp RNSTool::divide_and_round_q_last_ntt_inplace] -- reached from
 * SEAL_HEVM.cpp:283 (rescale_to_next) and, inlined, from switch_key_inplace's mod-down.
 * poly: [cnt][N], limb b modulo prime pidx[b]; the last limb is consumed, result in the first cnt-1.
 *   out_i = (x_i - ((x_last + floor(p/2)) mod p  reduced mod q_i) + (floor(p/2) mod q_i)) * p^{-1} mod q_i
 * ---------------------------------------------------------------------------------------------- */
void orc_divide_round_last(const orc_ctx *c, const int32_t *pidx, int cnt, u64 *poly)
{
    size_t N = c->N;
    int pl = pidx[cnt - 1];
    u64 p = c->mod[pl].q, half = p >> 1;
    u64 *last = poly + (size_t)(cnt - 1) * N;
    orc_ntt_inv(c, pl, last);
    for (size_t j = 0; j < N; j++) last[j] = addmod(last[j], half, p);
#pragma omp parallel for if (orc_threads > 1) schedule(dynamic)
    for (int b = 0; b < cnt - 1; b++) {
        int pi = pidx[b];
        const orc_mod *m = &c->mod[pi];
        u64 qi = m->q;
        u64 neg_half = qi - barrett64(half, m);
        u64 inv_p = invmod_prime(p % qi, qi);
        u64 *tmp = (u64 *)malloc(N * 8);
        for (size_t j = 0; j < N; j++) tmp[j] = addmod(barrett64(last[j], m), neg_half % qi, qi);
        orc_ntt_fwd(c, pi, tmp);
        u64 *x = poly + (size_t)b * N;
        for (size_t j = 0; j < N; j++) x[j] = mulmod(submod(x[j], tmp[j], qi), inv_p, m);
        free(tmp);
    }
}

/* Evaluator::rescale_to_next on one polynomial at level ell -> level ell-1 (SEAL_HEVM.cpp:283).
 * in: [ell][N] (not modified), out: [ell-1][N]. */
void orc_rescale_poly(const orc_ctx *c, int ell, const u64 *in, u64 *out)
{
    size_t N = c->N;
    int32_t pidx[ORC_MAX_PRIMES];
    for (int i = 0; i < ell; i++) pidx[i] = i;
    u64 *tmp = (u64 *)malloc((size_t)ell * N * 8);
    memcpy(tmp, in, (size_t)ell * N * 8);
    orc_divide_round_last(c, pidx, ell, tmp);
    memcpy(out, tmp, (size_t)(ell - 1) * N * 8);
    free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * Hybrid key switching with one special prime, one RNS prime per digit
 * [SEAL-upstream evaluator.cpp Evaluator::switch_key_inplace] -- reached from SEAL_HEVM.cpp:273
 * (rotate_vector -> apply_galois_inplace) and :316 (relinearize_inplace).
 *   target : [ell][N]  NTT form
 *   key    : [K-1][2][K][N]
 *   out0/out1 : [ell][N]; the switched pair is ADDED into them.
 * Steps follow SEAL: (1) iNTT the ell target limbs; (2) for every output modulus I in {q_0..q_{ell-1}, P}
 * and digit J: reduce digit J into modulus I (only if q_J > q_I), forward NTT (reuse the NTT-form limb
 * when I == J), multiply-accumulate with key[J][.][key_index(I)] in 128-bit lazy accumulators;
 * (3) mod-down by P with rounding, add into the ciphertext.
 * ---------------------------------------------------------------------------------------------- */
void orc_keyswitch(const orc_ctx *c, int ell, const u64 *target, const u64 *key, u64 *out0, u64 *out1)
{
    size_t N = c->N;
    int K = c->K, sp = K - 1;
    size_t key_poly = (size_t)K * N, key_digit = 2 * key_poly;
    u64 *t_target = (u64 *)malloc((size_t)ell * N * 8);
    u64 *prod = (u64 *)malloc((size_t)2 * (ell + 1) * N * 8); /* [2][ell+1][N] */
    memcpy(t_target, target, (size_t)ell * N * 8);
#pragma omp parallel for if (orc_threads > 1) schedule(dynamic)
    for (int j = 0; j < ell; j++) orc_ntt_inv(c, j, t_target + (size_t)j * N);

#pragma omp parallel for if (orc_threads > 1) schedule(dynamic)
    for (int I = 0; I <= ell; I++) {
        int ki = (I == ell) ? sp : I;
        const orc_mod *m = &c->mod[ki];
        /* per-modulus scratch (the serial build reuses one pair; a thread needs its own) */
        u64 *t_ntt = (u64 *)malloc(N * 8);
        u128 *acc = (u128 *)malloc((size_t)2 * N * sizeof(u128));
        memset(acc, 0, (size_t)2 * N * sizeof(u128));
        for (int J = 0; J < ell; J++) {
            const u64 *operand;
            if (I == J)
                operand = target + (size_t)J * N;
            else {
                const u64 *src = t_target + (size_t)J * N;
                if (c->mod[J].q <= m->q)
                    memcpy(t_ntt, src, N * 8);
                else
                    for (size_t n = 0; n < N; n++) t_ntt[n] = barrett64(src[n], m);
                orc_ntt_fwd(c, ki, t_ntt);
                operand = t_ntt;
            }
            const u64 *k0 = key + (size_t)J * key_digit + (size_t)ki * N;
            const u64 *k1 = k0 + key_poly;
            for (size_t n = 0; n < N; n++) {
                acc[n] += (u128)operand[n] * k0[n];
                acc[N + n] += (u128)operand[n] * k1[n];
            }
        }
        for (int kc = 0; kc < 2; kc++) {
            u64 *dst = prod + ((size_t)kc * (ell + 1) + I) * N;
            for (size_t n = 0; n < N; n++) dst[n] = barrett128(acc[(size_t)kc * N + n], m);
        }
        free(t_ntt);
        free(acc);
    }
    /* mod-down by the special prime (the CKKS branch of switch_key_inplace) */
    u64 P = c->mod[sp].q, half = P >> 1;
    for (int kc = 0; kc < 2; kc++) {
        u64 *pp = prod + (size_t)kc * (ell + 1) * N;
        u64 *t_last = pp + (size_t)ell * N;
        u64 *out = kc ? out1 : out0;
        orc_ntt_inv(c, sp, t_last);
        for (size_t n = 0; n < N; n++) t_last[n] = addmod(t_last[n], half, P);
#pragma omp parallel for if (orc_threads > 1) schedule(dynamic)
        for (int i = 0; i < ell; i++) {
            const orc_mod *m = &c->mod[i];
            u64 qi = m->q;
            u64 fix = qi - barrett64(half, m);
            u64 inv_p = invmod_prime(P % qi, qi);
            u64 *t_ntt = (u64 *)malloc(N * 8);
            for (size_t n = 0; n < N; n++) t_ntt[n] = addmod(barrett64(t_last[n], m), fix % qi, qi);
            orc_ntt_fwd(c, i, t_ntt);
            u64 *x = pp + (size_t)i * N;
            for (size_t n = 0; n < N; n++) {
                u64 v = mulmod(submod(x[n], t_ntt[n], qi), inv_p, m);
                out[(size_t)i * N + n] = addmod(out[(size_t)i * N + n], v, qi);
            }
            free(t_ntt);
        }
    }
    free(t_target);
    free(prod);
}

/* Closed form used to cross-check orc_keyswitch's digit products without lazy tricks:
 * acc[I][kc][n] = sum_J NTT_I(d_J mod q_I)[n] * key[J][kc][I][n] mod q_I, everything with %.
 * out: [2][ell+1][N]. */
void orc_keyswitch_inner_simple(const orc_ctx *c, int ell, const u64 *target, const u64 *key, u64 *out)
{
    size_t N = c->N;
    int K = c->K, sp = K - 1;
    size_t key_poly = (size_t)K * N, key_digit = 2 * key_poly;
    u64 *d = (u64 *)malloc((size_t)ell * N * 8);
    u64 *t = (u64 *)malloc(N * 8);
    memcpy(d, target, (size_t)ell * N * 8);
    for (int j = 0; j < ell; j++) orc_ntt_inv_simple(c, j, d + (size_t)j * N);
    memset(out, 0, (size_t)2 * (ell + 1) * N * 8);
    for (int I = 0; I <= ell; I++) {
        int ki = (I == ell) ? sp : I;
        u64 q = c->mod[ki].q;
        for (int J = 0; J < ell; J++) {
            for (size_t n = 0; n < N; n++) t[n] = d[(size_t)J * N + n] % q;
            orc_ntt_fwd_simple(c, ki, t);
            for (int kc = 0; kc < 2; kc++) {
                const u64 *kk = key + (size_t)J * key_digit + (size_t)kc * key_poly + (size_t)ki * N;
                u64 *o = out + ((size_t)kc * (ell + 1) + I) * N;
                for (size_t n = 0; n < N; n++) o[n] = (o[n] + mulmod_simple(t[n], kk[n], q)) % q;
            }
        }
    }
    free(d);
    free(t);
}

/* ------------------------------------------------------------------------------------------------
 * EXTENSION (not SEAL, not in the reference's SEAL runtime): hybrid key switching with GROUPED digits -- dnum = ceil(L / alpha)
 * digits of alpha data primes each and ks special primes (Han-Ki 2020; what the reference's HEaaN runtime does inside its closed
 * library: HEAAN_HEVM.cpp:124-141 keygen, :386-399 bootstrap at 29 levels).  The published algorithm, restated on canonical residues
 * with the INTEGER fast base conversion (no floating-point correction), so that every implementation agrees bit for bit:
 *   digit g of the target = its residues modulo S_g = {q_i : g alpha <= i < min((g+1) alpha, ell)} (coefficient domain), raised to
 *   every other modulus m of {q_0..q_{ell-1}, p_0..p_{ks-1}} as   sum_{i in S_g} [x_i (Q_g/q_i)^{-1}]_{q_i} (Q_g/q_i)  mod m
 *   (= x + u Q_g, 0 <= u < |S_g|: the overshoot only multiplies the key's error term); inner products with key[g];
 *   mod-down by P = p_0 ... p_{ks-1} with rounding: r = [acc + floor(P/2)]_P converted the same way to q_i, out_i = (acc_i -
 *   (conv_i - floor(P/2) mod q_i)) P^{-1} mod q_i  (= round(acc / P) - u', 0 <= u' < ks).
 * With ks = alpha = 1 every step degenerates to SEAL's switch_key_inplace above (tested: identical limbs).
 * key: [dnum][2][K][N], digit g = (-(a s + e) + [i in group g] (P mod q_i) s'_i , a).
 * ---------------------------------------------------------------------------------------------- */
int orc_set_hybrid(orc_ctx *c, int ks, int alpha)
{
    if (ks < 1 || alpha < 1 || alpha > ks || ks >= c->K) return -1; /* P must cover a digit: alpha <= ks */
    c->ks = ks;
    c->alpha = alpha;
    return 0;
}
int orc_hybrid_dnum(const orc_ctx *c) { return (c->K - c->ks + c->alpha - 1) / c->alpha; }

static u64 prod_mod_except(const orc_ctx *c, int lo, int hi, int skip, u64 m)
{ /* product of primes lo..hi-1 except `skip`, modulo m */
    u64 r = 1 % m;
    for (int t = lo; t < hi; t++)
        if (t != skip) r = mulmod_simple(r, c->mod[t].q % m, m);
    return r;
}

static void keyswitch_hybrid_perm(const orc_ctx *c, int ell, const u64 *target, const u64 *key, u64 *out0, u64 *out1, const uint32_t *perm);
static void hybrid_accumulate(const orc_ctx *c, int ell, const u64 *target, const u64 *key, const uint32_t *perm, u64 *prod);
static void hybrid_moddown(const orc_ctx *c, int ell, u64 *prod, u64 *out0, u64 *out1);
void orc_keyswitch_hybrid(const orc_ctx *c, int ell, const u64 *target, const u64 *key, u64 *out0, u64 *out1)
{
    keyswitch_hybrid_perm(c, ell, target, key, out0, out1, NULL);
}
/* The key-switch half of a ROTATION in grouped-digit mode: (out0, out1) += KS(galois(c1)) with the digits taken BEFORE the automorphism --
 * decompose c1 (inverse NTT, mod-up, NTT), then apply the Galois permutation to every raised limb in the NTT domain (a ring automorphism
 * maps a valid decomposition of c1 to a valid decomposition of galois(c1)).  That order is what lets rotations of ONE ciphertext share the
 * decomposition ("hoisting": bootstrapping's baby steps, a convolution's taps); it is this runtime's definition of a grouped-digit rotation
 * whether or not anything is shared, so hoisted and un-hoisted executions agree bit for bit.  (SEAL's scheme rotates first; this mode is
 * not SEAL's.) */
void orc_rotate_ks_hybrid(const orc_ctx *c, int ell, const u64 *c1, uint32_t elt, const u64 *key, u64 *out0, u64 *out1)
{
    uint32_t *perm = (uint32_t *)malloc(c->N * sizeof(uint32_t));
    orc_galois_table(c, elt, perm);
    keyswitch_hybrid_perm(c, ell, c1, key, out0, out1, perm);
    free(perm);
}
/* EXTENSION of the extension (the GPU VM's option hyb_lazy_sum): a SUM of grouped-digit rotations with ONE mod-down -- "double hoisting"
 * (Bossuat, Mouchet, Troncoso-Pastoriza, Hubaux 2021, section 5: the giant steps of a BSGS matrix-vector product).  The inner products of
 * every rotation of the sum are added in the raised basis, then divided by P once:  sum_r galois_r(c0_r) + moddown(sum_r acc_r).  One
 * rounding instead of one per rotation, so the limbs differ from the sum of orc_rotate_ks_hybrid results (by the roundings' difference, a
 * few units) and the noise is, if anything, smaller.  orc_rotate_acc_hybrid adds one rotation's inner products to prod [2][ell + ks][N];
 * orc_moddown_hybrid adds moddown(prod) to (out0, out1) and destroys prod.  A "sum" of one rotation is orc_rotate_ks_hybrid exactly. */
void orc_rotate_acc_hybrid(const orc_ctx *c, int ell, const u64 *c1, uint32_t elt, const u64 *key, u64 *prod)
{
    uint32_t *perm = (uint32_t *)malloc(c->N * sizeof(uint32_t));
    orc_galois_table(c, elt, perm);
    hybrid_accumulate(c, ell, c1, key, perm, prod);
    free(perm);
}
void orc_moddown_hybrid(const orc_ctx *c, int ell, u64 *prod, u64 *out0, u64 *out1) { hybrid_moddown(c, ell, prod, out0, out1); }

static void keyswitch_hybrid_perm(const orc_ctx *c, int ell, const u64 *target, const u64 *key, u64 *out0, u64 *out1, const uint32_t *perm)
{
    u64 *prod = (u64 *)calloc((size_t)2 * (ell + c->ks) * c->N, 8); /* [2][ell + ks][N], canonical running sums */
    hybrid_accumulate(c, ell, target, key, perm, prod);
    hybrid_moddown(c, ell, prod, out0, out1);
    free(prod);
}
/* prod [2][ell + ks][N] += the inner products of target's digits (read through perm when given) with the key */
static void hybrid_accumulate(const orc_ctx *c, int ell, const u64 *target, const u64 *key, const uint32_t *perm, u64 *prod)
{
    size_t N = c->N;
    int K = c->K, ks = c->ks, al = c->alpha, L = K - ks, M = ell + ks;
    int G = (ell + al - 1) / al;
    size_t key_poly = (size_t)K * N, key_digit = 2 * key_poly;
    u64 *coef = (u64 *)malloc((size_t)ell * N * 8);
    u64 *y = (u64 *)malloc((size_t)al * N * 8);
    u64 *t = (u64 *)malloc(N * 8);
    memcpy(coef, target, (size_t)ell * N * 8);
    for (int j = 0; j < ell; j++) orc_ntt_inv(c, j, coef + (size_t)j * N);
    for (int g = 0; g < G; g++) {
        int lo = g * al, hi = lo + al < ell ? lo + al : ell;
        for (int i = lo; i < hi; i++) { /* y_i = [x_i (Q_g/q_i)^{-1}]_{q_i} */
            const orc_mod *mi = &c->mod[i];
            u64 inv = invmod_prime(prod_mod_except(c, lo, hi, i, mi->q), mi->q);
            for (size_t n = 0; n < N; n++) y[(size_t)(i - lo) * N + n] = mulmod(coef[(size_t)i * N + n], inv, mi);
        }
        for (int I = 0; I < M; I++) {
            int pm = I < ell ? I : L + (I - ell);
            const orc_mod *m = &c->mod[pm];
            const u64 *operand;
            if (I >= lo && I < hi)
                operand = target + (size_t)I * N; /* a modulus of the digit's own group: the NTT-form limb itself */
            else {
                u64 w[ORC_MAX_PRIMES];
                for (int i = lo; i < hi; i++) w[i - lo] = prod_mod_except(c, lo, hi, i, m->q);
                for (size_t n = 0; n < N; n++) {
                    u64 a = 0;
                    for (int i = lo; i < hi; i++) a = addmod(a, mulmod(barrett64(y[(size_t)(i - lo) * N + n], m), w[i - lo], m), m->q);
                    t[n] = a;
                }
                orc_ntt_fwd(c, pm, t);
                operand = t;
            }
            for (int kc = 0; kc < 2; kc++) {
                const u64 *kk = key + (size_t)g * key_digit + (size_t)kc * key_poly + (size_t)pm * N;
                u64 *o = prod + ((size_t)kc * M + I) * N;
                if (perm)
                    for (size_t n = 0; n < N; n++) o[n] = addmod(o[n], mulmod(operand[perm[n]], kk[n], m), m->q);
                else
                    for (size_t n = 0; n < N; n++) o[n] = addmod(o[n], mulmod(operand[n], kk[n], m), m->q);
            }
        }
    }
    free(coef);
    free(y);
    free(t);
}
/* (out0, out1) += prod / P, rounded (prod's special limbs are overwritten) */
static void hybrid_moddown(const orc_ctx *c, int ell, u64 *prod, u64 *out0, u64 *out1)
{
    size_t N = c->N;
    int K = c->K, ks = c->ks, L = K - ks, M = ell + ks;
    u64 *t = (u64 *)malloc(N * 8);
    /* mod-down by P with rounding */
    u64 *z = (u64 *)malloc((size_t)ks * N * 8);
    for (int kc = 0; kc < 2; kc++) {
        u64 *pp = prod + (size_t)kc * M * N;
        u64 *out = kc ? out1 : out0;
        for (int j = 0; j < ks; j++) {
            int pj = L + j;
            const orc_mod *m = &c->mod[pj];
            u64 *r = pp + (size_t)(ell + j) * N;
            orc_ntt_inv(c, pj, r);
            u64 Pm = prod_mod_except(c, L, K, -1, m->q); /* = 0 */
            (void)Pm;
            /* floor(P/2) mod p_j: P is odd and = 0 mod p_j, so floor(P/2) = (P - 1)/2 = -(1/2) mod p_j = (p_j - 1)/2 */
            u64 half = (m->q - 1) >> 1;
            u64 inv = invmod_prime(prod_mod_except(c, L, K, pj, m->q), m->q);
            for (size_t n = 0; n < N; n++) z[(size_t)j * N + n] = mulmod(addmod(r[n], half, m->q), inv, m);
        }
        for (int i = 0; i < ell; i++) {
            const orc_mod *m = &c->mod[i];
            u64 qi = m->q, w[ORC_MAX_PRIMES];
            for (int j = 0; j < ks; j++) w[j] = prod_mod_except(c, L, K, L + j, qi);
            u64 Pq = prod_mod_except(c, L, K, -1, qi);
            u64 inv2 = (qi + 1) >> 1;                                   /* 2^{-1} mod q_i */
            u64 half_q = mulmod(submod(Pq, 1 % qi, qi), inv2, m);       /* floor(P/2) = (P - 1)/2 mod q_i */
            u64 inv_p = invmod_prime(Pq, qi);
            for (size_t n = 0; n < N; n++) {
                u64 a = 0;
                for (int j = 0; j < ks; j++) a = addmod(a, mulmod(barrett64(z[(size_t)j * N + n], m), w[j], m), qi);
                t[n] = submod(a, half_q, qi);
            }
            orc_ntt_fwd(c, i, t);
            u64 *x = pp + (size_t)i * N;
            for (size_t n = 0; n < N; n++) {
                u64 v = mulmod(submod(x[n], t[n], qi), inv_p, m);
                out[(size_t)i * N + n] = addmod(out[(size_t)i * N + n], v, qi);
            }
        }
    }
    free(t);
    free(z);
}

/* ------------------------------------------------------------------------------------------------
 * CKKS encoder / decoder  [SEAL-upstream ckks.h/ckks.cpp CKKSEncoder::encode_internal /
 * decode_internal, dwthandler.h transform_from_rev / transform_to_rev]
 * -- reached from SEAL_HEVM.cpp:262 (encode), :331,:451 (decode).
 * ---------------------------------------------------------------------------------------------- */
static void fft_from_rev(const orc_ctx *c, double complex *v, double fix)
{
    size_t n = c->N;
    for (size_t m = n >> 1, gap = 1; m > 1; m >>= 1, gap <<= 1)
        for (size_t i = 0; i < m; i++) {
            double complex r = conj(c->croot[m + i]);
            double complex *x = v + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                double complex u = x[j], w = y[j];
                x[j] = u + w;
                y[j] = (u - w) * r;
            }
        }
    { /* last stage carries the scalar (scale / n) */
        size_t gap = n >> 1;
        double complex r = conj(c->croot[1]);
        double complex sr = r * fix;
        for (size_t j = 0; j < gap; j++) {
            double complex u = v[j], w = v[j + gap];
            v[j] = (u + w) * fix;
            v[j + gap] = (u - w) * sr;
        }
    }
}
static void fft_to_rev(const orc_ctx *c, double complex *v)
{
    size_t n = c->N;
    for (size_t m = 1, gap = n >> 1; m < n; m <<= 1, gap >>= 1)
        for (size_t i = 0; i < m; i++) {
            double complex r = c->croot[m + i];
            double complex *x = v + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                double complex u = x[j], w = y[j] * r;
                x[j] = u + w;
                y[j] = u - w;
            }
        }
}

/* values: nvals (<= N/2) real numbers (imaginary part 0 -- HEVM only ever encodes reals);
 * out: [ell][N] NTT form.  Returns 0, or -1 if a coefficient does not fit in 127 bits. */
static int encode_from_slots(const orc_ctx *c, double complex *cv, double scale, int ell, u64 *out);

int orc_encode(const orc_ctx *c, const double *values, size_t nvals, double scale, int ell, u64 *out)
{
    size_t N = c->N, slots = N >> 1;
    if (nvals > slots) return -1;
    double complex *cv = (double complex *)calloc(N, sizeof(double complex));
    for (size_t i = 0; i < nvals; i++) {
        cv[c->slot_map[i]] = values[i];
        cv[c->slot_map[slots + i]] = values[i]; /* conj of a real */
    }
    return encode_from_slots(c, cv, scale, ell, out);
}

/* complex slot values (extension opcode 16 of this repo's VM; CKKSEncoder::encode of a complex vector [SEAL-upstream]) */
int orc_encode_complex(const orc_ctx *c, const double *re, const double *im, size_t nvals, double scale, int ell, u64 *out)
{
    size_t N = c->N, slots = N >> 1;
    if (nvals > slots) return -1;
    double complex *cv = (double complex *)calloc(N, sizeof(double complex));
    for (size_t i = 0; i < nvals; i++) {
        cv[c->slot_map[i]] = CMPLX(re[i], im[i]);
        cv[c->slot_map[slots + i]] = CMPLX(re[i], -im[i]);
    }
    return encode_from_slots(c, cv, scale, ell, out);
}

static int encode_from_slots(const orc_ctx *c, double complex *cv, double scale, int ell, u64 *out)
{
    size_t N = c->N;
    fft_from_rev(c, cv, scale / (double)N);
    int rc = 0;
    for (size_t j = 0; j < N; j++) {
        double coeffd = round(creal(cv[j]));
        int neg = signbit(coeffd) != 0;
        double mag = fabs(coeffd);
        if (mag >= 0x1p127) {
            rc = -1;
            break;
        }
        u128 u;
        if (mag < 0x1p63)
            u = (u64)mag;
        else {
            int e;
            double fr = frexp(mag, &e); /* mag = fr * 2^e, fr in [0.5,1) */
            u64 mant = (u64)ldexp(fr, 53);
            u = ((u128)mant) << (e - 53);
        }
        for (int i = 0; i < ell; i++) {
            u64 q = c->mod[i].q;
            u64 r = (u64)(u % q);
            out[(size_t)i * N + j] = neg ? negmod(r, q) : r;
        }
    }
    free(cv);
    if (rc) return rc;
    for (int i = 0; i < ell; i++) orc_ntt_fwd(c, i, out + (size_t)i * N);
    return 0;
}

/* multi-precision helpers (little-endian u64 words) */
static void mp_mul_u64_add(u64 *acc, const u64 *a, int words, u64 b)
{ /* acc += a*b (acc has `words` words; overflow beyond is impossible by construction) */
    u128 carry = 0;
    for (int i = 0; i < words; i++) {
        u128 t = (u128)a[i] * b + acc[i] + (u64)carry;
        acc[i] = (u64)t;
        carry = t >> 64;
    }
}
static int mp_cmp(const u64 *a, const u64 *b, int words)
{
    for (int i = words - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return -1;
    }
    return 0;
}

/* plain: [ell][N] NTT form (not modified). out: N/2 doubles (real parts), as SEAL_HEVM.cpp:450-454. */
void orc_decode(const orc_ctx *c, const u64 *plain, int ell, double scale, double *out)
{
    size_t N = c->N, slots = N >> 1;
    int W = ell;
    u64 *coef = (u64 *)malloc((size_t)ell * N * 8);
    memcpy(coef, plain, (size_t)ell * N * 8);
    for (int i = 0; i < ell; i++) orc_ntt_inv(c, i, coef + (size_t)i * N);
    /* Garner mixed-radix constants: M[k] = prod_{i<k} q_i (multi-word), inv[k] = M[k]^{-1} mod q_k */
    u64 *M = (u64 *)calloc((size_t)(ell + 1) * W, 8);
    M[0] = 1;
    for (int k = 1; k <= ell; k++) mp_mul_u64_add(M + (size_t)k * W, M + (size_t)(k - 1) * W, W, c->mod[k - 1].q);
    u64 Mmod[ORC_MAX_PRIMES][ORC_MAX_PRIMES]; /* M[i] mod q_k */
    u64 inv[ORC_MAX_PRIMES];
    for (int k = 0; k < ell; k++) {
        u64 qk = c->mod[k].q;
        u64 acc = 1;
        for (int i = 0; i <= k; i++) {
            Mmod[i][k] = acc;
            if (i < k) acc = mulmod_simple(acc, c->mod[i].q % qk, qk);
        }
        inv[k] = invmod_prime(Mmod[k][k], qk);
    }
    u64 *Q = M + (size_t)ell * W; /* total modulus */
    u64 half[ORC_MAX_PRIMES];     /* upper_half_threshold = (Q+1)>>1 */
    {
        u64 carry = 1;
        u64 tmp[ORC_MAX_PRIMES];
        for (int i = 0; i < W; i++) {
            tmp[i] = Q[i] + carry;
            carry = (carry && tmp[i] == 0) ? 1 : 0;
        }
        for (int i = 0; i < W; i++) half[i] = (tmp[i] >> 1) | ((i + 1 < W ? tmp[i + 1] : carry) << 63);
    }
    double complex *res = (double complex *)malloc(N * sizeof(double complex));
    double inv_scale = 1.0 / scale, two64 = 0x1p64;
    u64 v[ORC_MAX_PRIMES], X[ORC_MAX_PRIMES];
    for (size_t n = 0; n < N; n++) {
        for (int k = 0; k < ell; k++) {
            u64 qk = c->mod[k].q, s = 0;
            for (int i = 0; i < k; i++) s = (s + mulmod_simple(v[i] % qk, Mmod[i][k], qk)) % qk;
            u64 xk = coef[(size_t)k * N + n];
            v[k] = mulmod_simple((xk + qk - s) % qk, inv[k], qk);
        }
        memset(X, 0, sizeof(u64) * W);
        for (int k = 0; k < ell; k++) mp_mul_u64_add(X, M + (size_t)k * W, W, v[k]);
        double r = 0.0, sc = inv_scale;
        if (mp_cmp(X, half, W) >= 0) {
            for (int j = 0; j < W; j++, sc *= two64) {
                if (X[j] > Q[j]) {
                    u64 d = X[j] - Q[j];
                    r += d ? (double)d * sc : 0.0;
                } else {
                    u64 d = Q[j] - X[j];
                    r -= d ? (double)d * sc : 0.0;
                }
            }
        } else {
            for (int j = 0; j < W; j++, sc *= two64) r += X[j] ? (double)X[j] * sc : 0.0;
        }
        res[n] = r;
    }
    fft_to_rev(c, res);
    for (size_t i = 0; i < slots; i++) out[i] = creal(res[c->slot_map[i]]);
    free(res);
    free(M);
    free(coef);
}

/* ------------------------------------------------------------------------------------------------
 * Sampling, keys, encrypt, decrypt  [SEAL-upstream rlwe.cpp sample_poly_ternary / sample_poly_cbd /
 * sample_poly_uniform / encrypt_zero_asymmetric / encrypt_zero_symmetric, keygenerator.cpp,
 * encryptor.cpp, decryptor.cpp] -- reached from SEAL_HEVM.cpp:60-83 (keygen), :444 (encrypt),
 * :449 (decrypt).  SEAL draws randomness from Blake2xb/Shake256; this oracle uses splitmix64, so key
 * material is distribution-equivalent, not byte-identical.
 * ---------------------------------------------------------------------------------------------- */
static inline u64 splitmix64(u64 *s)
{
    u64 z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
u64 orc_splitmix64(u64 *s) { return splitmix64(s); }

/* signed small polynomial -> RNS, coefficient domain, primes 0..cnt-1 */
static void small_to_rns(const orc_ctx *c, const int *s, int cnt, u64 *out)
{
    for (int i = 0; i < cnt; i++) {
        u64 q = c->mod[i].q;
        for (size_t j = 0; j < c->N; j++) out[(size_t)i * c->N + j] = s[j] < 0 ? q - (u64)(-s[j]) : (u64)s[j];
    }
}
static void sample_ternary(const orc_ctx *c, u64 *seed, int *out)
{
    for (size_t j = 0; j < c->N; j++) out[j] = (int)(splitmix64(seed) % 3) - 1;
}
static void sample_cbd(const orc_ctx *c, u64 *seed, int *out)
{ /* hamming(21 bits) - hamming(21 bits): sigma = sqrt(21/2) ~ 3.24 */
    for (size_t j = 0; j < c->N; j++) {
        u64 r = splitmix64(seed);
        out[j] = __builtin_popcountll(r & 0x1FFFFF) - __builtin_popcountll((r >> 21) & 0x1FFFFF);
    }
}
static void sample_uniform(const orc_ctx *c, u64 *seed, int cnt, u64 *out)
{
    for (int i = 0; i < cnt; i++) {
        u64 q = c->mod[i].q;
        u64 max_multiple = UINT64_MAX - (UINT64_MAX % q) - 1;
        for (size_t j = 0; j < c->N; j++) {
            u64 r;
            do r = splitmix64(seed);
            while (r >= max_multiple);
            out[(size_t)i * c->N + j] = r % q;
        }
    }
}

/* secret key, NTT form over all K primes: [K][N] */
void orc_gen_secret(const orc_ctx *c, u64 *seed, u64 *sk)
{
    int *s = (int *)malloc(c->N * sizeof(int));
    sample_ternary(c, seed, s);
    small_to_rns(c, s, c->K, sk);
    for (int i = 0; i < c->K; i++) orc_ntt_fwd(c, i, sk + (size_t)i * c->N);
    free(s);
}

/* (c0, c1) = (-(a s + e), a) over primes 0..cnt-1, NTT form; out: [2][cnt][N].
 * sk is [K][N] (limb stride N, first cnt limbs used). */
static void encrypt_zero_symmetric(const orc_ctx *c, const u64 *sk, int cnt, u64 *seed, u64 *out)
{
    size_t N = c->N;
    int *e = (int *)malloc(N * sizeof(int));
    u64 *c0 = out, *c1 = out + (size_t)cnt * N;
    sample_uniform(c, seed, cnt, c1);
    sample_cbd(c, seed, e);
    small_to_rns(c, e, cnt, c0);
    for (int i = 0; i < cnt; i++) {
        const orc_mod *m = &c->mod[i];
        orc_ntt_fwd(c, i, c0 + (size_t)i * N);
        for (size_t j = 0; j < N; j++) {
            size_t k = (size_t)i * N + j;
            c0[k] = negmod(addmod(mulmod(c1[k], sk[k], m), c0[k], m->q), m->q);
        }
    }
    free(e);
}

/* public key at key level: [2][K][N] */
void orc_gen_public(const orc_ctx *c, const u64 *sk, u64 *seed, u64 *pk)
{
    encrypt_zero_symmetric(c, sk, c->K, seed, pk);
}

/* key-switch key from `new_key` ([K][N], NTT form) to sk: [K-1][2][K][N];
 * digit j: encrypt_zero_symmetric at key level, then c0[limb j] += (P mod q_j) * new_key[limb j]
 * [SEAL-upstream KeyGenerator::generate_one_kswitch_key]. */
void orc_gen_kswitch(const orc_ctx *c, const u64 *sk, const u64 *new_key, u64 *seed, u64 *ksk)
{
    size_t N = c->N;
    int K = c->K;
    u64 P = c->mod[K - 1].q;
    for (int j = 0; j < K - 1; j++) {
        u64 *dj = ksk + (size_t)j * 2 * K * N;
        encrypt_zero_symmetric(c, sk, K, seed, dj);
        const orc_mod *m = &c->mod[j];
        u64 factor = barrett64(P, m);
        for (size_t n = 0; n < N; n++) {
            size_t k = (size_t)j * N + n;
            dj[k] = addmod(dj[k], mulmod(new_key[k], factor, m), m->q);
        }
    }
}

/* grouped-digit key (extension, see orc_keyswitch_hybrid): [dnum][2][K][N] */
void orc_gen_kswitch_hybrid(const orc_ctx *c, const u64 *sk, const u64 *new_key, u64 *seed, u64 *ksk)
{
    size_t N = c->N;
    int K = c->K, L = K - c->ks, dnum = orc_hybrid_dnum(c);
    for (int g = 0; g < dnum; g++) {
        u64 *dg = ksk + (size_t)g * 2 * K * N;
        encrypt_zero_symmetric(c, sk, K, seed, dg);
        int lo = g * c->alpha, hi = lo + c->alpha < L ? lo + c->alpha : L;
        for (int i = lo; i < hi; i++) {
            const orc_mod *m = &c->mod[i];
            u64 factor = prod_mod_except(c, L, K, -1, m->q); /* P mod q_i */
            for (size_t n = 0; n < N; n++) {
                size_t k = (size_t)i * N + n;
                dg[k] = addmod(dg[k], mulmod(new_key[k], factor, m), m->q);
            }
        }
    }
}

/* Encryptor::encrypt at level ell (primes 0..ell-1) of an NTT-form plaintext [ell][N]:
 * encrypt zero under the public key with primes 0..ell (the "previous" context), divide-and-round
 * by prime ell, add the plaintext to c0.  pk: [2][K][N].  out: [2][ell][N].  Requires ell < K. */
void orc_encrypt(const orc_ctx *c, const u64 *pk, const u64 *plain, int ell, u64 *seed, u64 *out)
{
    size_t N = c->N;
    int cnt = ell + 1, K = c->K;
    int *u = (int *)malloc(N * sizeof(int)), *e = (int *)malloc(N * sizeof(int));
    u64 *un = (u64 *)malloc((size_t)cnt * N * 8);
    u64 *tmp = (u64 *)malloc((size_t)cnt * N * 8);
    int32_t pidx[ORC_MAX_PRIMES];
    for (int i = 0; i < cnt; i++) pidx[i] = i;
    sample_ternary(c, seed, u);
    small_to_rns(c, u, cnt, un);
    for (int i = 0; i < cnt; i++) orc_ntt_fwd(c, i, un + (size_t)i * N);
    for (int kc = 0; kc < 2; kc++) {
        sample_cbd(c, seed, e);
        small_to_rns(c, e, cnt, tmp);
        for (int i = 0; i < cnt; i++) {
            const orc_mod *m = &c->mod[i];
            orc_ntt_fwd(c, i, tmp + (size_t)i * N);
            const u64 *pkl = pk + ((size_t)kc * K + i) * N;
            for (size_t j = 0; j < N; j++) {
                size_t k = (size_t)i * N + j;
                tmp[k] = addmod(mulmod(pkl[j], un[k], m), tmp[k], m->q);
            }
        }
        orc_divide_round_last(c, pidx, cnt, tmp);
        memcpy(out + (size_t)kc * ell * N, tmp, (size_t)ell * N * 8);
    }
    for (int i = 0; i < ell; i++)
        for (size_t j = 0; j < N; j++) {
            size_t k = (size_t)i * N + j;
            out[k] = addmod(out[k], plain[k], c->mod[i].q);
        }
    free(u);
    free(e);
    free(un);
    free(tmp);
}

/* Decryptor::decrypt (size-2 ciphertext): plain = c0 + c1*s, NTT form.  ct: [2][ell][N]. */
void orc_decrypt(const orc_ctx *c, const u64 *sk, const u64 *ct, int ell, u64 *plain)
{
    size_t N = c->N;
    for (int i = 0; i < ell; i++) {
        const orc_mod *m = &c->mod[i];
        for (size_t j = 0; j < N; j++) {
            size_t k = (size_t)i * N + j;
            plain[k] = addmod(ct[k], mulmod(ct[(size_t)ell * N + k], sk[k], m), m->q);
        }
    }
}
