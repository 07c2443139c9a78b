// Batched negacyclic NTT / inverse NTT over RNS limbs: two launches per transform (see ntt_tile.hpp).
// Replaces SEAL's ntt_negacyclic_harvey / inverse_ntt_negacyclic_harvey [SEAL-upstream ntt.cpp], which
// SEAL_HEVM.cpp reaches through every rotate (:273), rescale (:283), relinearize (:316) and encode (:262).
#include "kernels.hpp"
#include "ntt_tile.hpp"

namespace dacapo {

template <int K, int LOGE, bool COLS, bool INV, bool CANON>
__global__ __launch_bounds__(kTileThreads) void ntt_phase_kernel(u64 *__restrict__ data, long limb_stride,
                                                                  const int *__restrict__ prime_idx, int prime_base,
                                                                  int prime_period, const DModulus *__restrict__ mods,
                                                                  const u64 *__restrict__ tw, int logN, int prime_major)
{
    __shared__ __attribute__((aligned(16))) u64 lds[TileGeo<LOGE>::LDS_ELEMS];
    // prime_major (ROWS phases on large rings, option rows_prime_major): row y of the grid is limb (y mod batches) * period + y / batches --
    // the `batches` limbs that share a prime are adjacent in launch order.  A ROWS tile's twiddles are its own 2^K-entry slice of the prime's
    // table, so a limb's transform reads as many twiddle bytes as data bytes; at N = 2^17 a table is 1 MiB per prime and 40 of them do not
    // stay in a 4 MiB L2 from one limb of a prime to the next one `period` limbs later.  Adjacent, tile t of the next limb (same XCD: the
    // tile count is a multiple of 8) finds them there.
    int limb = blockIdx.y;
    if (prime_major) {
        const int batches = gridDim.y / prime_period;
        limb = (int)(blockIdx.y % batches) * prime_period + (int)(blockIdx.y / batches);
    }
    const int p = prime_idx ? prime_idx[limb % prime_period] : prime_base + (limb % prime_period);
    u64 *a = data + (long)limb * limb_stride;
    const DModulus M = mods[p];
    ntt_tile<K, LOGE, COLS, INV, CANON>(
        M, tw + ((size_t)p << logN), logN, blockIdx.x, [=](int i) { return a[i]; }, [=](int i, u64 v) { a[i] = v; }, lds);
}

template <int K, int LOGE, bool COLS, bool INV, bool CANON>
static void launch_phase(const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                         int prime_period, hipStream_t s, const DModulus *mods)
{
    dim3 grid((unsigned)(c.N >> TileGeo<LOGE>::LOG), (unsigned)count);
    DC_LAUNCH((ntt_phase_kernel<K, LOGE, COLS, INV, CANON>), grid, dim3(kTileThreads), 0, s, data, limb_stride,
                       d_prime_idx, prime_base, prime_period, mods ? mods : c.d_mods, INV ? c.d_itw : c.d_tw, c.logN,
                       (int)(!COLS && rows_prime_major(c) && count > prime_period && count % prime_period == 0));
}

template <bool COLS, bool INV, bool CANON>
static void launch_phase_k(int K, const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx,
                           int prime_base, int prime_period, hipStream_t s, const DModulus *mods = nullptr)
{
    const bool small = use_small_tiles(c.N, count), tiny = use_tiny_tiles(c.N, count);
#define DC_PHASE(KK)                                                                                                           \
    case KK:                                                                                                                   \
        if (tiny)                                                                                                              \
            launch_phase<KK, 1, COLS, INV, CANON>(c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s, mods); \
        else if (small)                                                                                                        \
            launch_phase<KK, 2, COLS, INV, CANON>(c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s, mods); \
        else                                                                                                                   \
            launch_phase<KK, 3, COLS, INV, CANON>(c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s, mods); \
        break;
    switch (K) {
        DC_PHASE(6)
        DC_PHASE(7)
        DC_PHASE(8)
        DC_PHASE(9)
    default: fprintf(stderr, "[dacapo_amd] unsupported NTT phase size 2^%d\n", K); abort();
    }
#undef DC_PHASE
}

void launch_ntt_rows_fwd(const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                         int prime_period, hipStream_t s)
{
    if (count <= 0) return;
    if (prime_period <= 0) prime_period = 1 << 30;
    launch_phase_k<false, false, true>(c.k2, c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s);
}

void launch_ntt_cols_fwd(const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                         int prime_period, hipStream_t s)
{
    if (count <= 0) return;
    if (prime_period <= 0) prime_period = 1 << 30;
    launch_phase_k<true, false, false>(c.k1, c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s);
}

void launch_ntt_cols_inv(const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                         int prime_period, hipStream_t s, const DModulus *mods)
{
    if (count <= 0) return;
    if (prime_period <= 0) prime_period = 1 << 30;
    launch_phase_k<true, true, true>(c.k1, c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s, mods);
}

void launch_ntt_rows_inv(const Context &c, u64 *data, long limb_stride, int count, const int *d_prime_idx, int prime_base,
                         int prime_period, hipStream_t s)
{
    if (count <= 0) return;
    if (prime_period <= 0) prime_period = 1 << 30;
    launch_phase_k<false, true, false>(c.k2, c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s);
}

void launch_ntt(const Context &c, bool inverse, u64 *data, long limb_stride, int count, const int *d_prime_idx,
                int prime_base, int prime_period, hipStream_t s)
{
    if (ntt_full_supported(c) && ntt_full_pays(inverse, count))
        launch_ntt_full(c, inverse, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s);
    else
        launch_ntt_two_phase(c, inverse, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s);
}

void launch_ntt_two_phase(const Context &c, bool inverse, u64 *data, long limb_stride, int count, const int *d_prime_idx,
                          int prime_base, int prime_period, hipStream_t s)
{
    if (count <= 0) return;
    if (prime_period <= 0) prime_period = 1 << 30;
    if (!inverse) {
        launch_phase_k<true, false, false>(c.k1, c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s);
        launch_phase_k<false, false, true>(c.k2, c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s);
    } else {
        launch_phase_k<false, true, false>(c.k2, c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s);
        launch_phase_k<true, true, true>(c.k1, c, data, limb_stride, count, d_prime_idx, prime_base, prime_period, s);
    }
}

} // namespace dacapo
