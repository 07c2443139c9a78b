// Host-side CKKS context of the MI355X HEVM runtime: prime chain, roots, twiddle tables resident in HBM,
// per-level scratch for key switching.  Mirrors what SEAL builds in SEALContext / NTTTables for
// SEAL_HEVM.cpp:46-59,93-99 (parameters N = 2^15, CoeffModulus::Create(N, {60 x 14})).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "modarith.hpp"
#include "options.hpp"

namespace dacapo {

#define DC_HIP_CHECK(expr)                                                                                     \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            fprintf(stderr, "[dacapo_amd] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, __LINE__, \
                    hipGetErrorString(e_));                                                                    \
            abort();                                                                                           \
        }                                                                                                      \
    } while (0)

// Every kernel launch of the library goes through this: hipLaunchKernelGGL + a check of the launch itself (a grid or block the device
// refuses, too much LDS or too many registers for the block size, an invalid function) at the launch site, not at some later synchronise.
// hipGetLastError is a host-side read of the thread's last error: free outside a capture, and inside a stream capture it still reports
// what the record-time validation of the launch found.  The thread's error state is read (and thereby cleared) BEFORE the launch too: on
// runtimes where the last error is sticky, an unrelated earlier failure on this thread -- PyTorch and RCCL share it in bench.py and
// tests/test_gpu_rccl.py, and a benign hipErrorInvalidValue from a pointer-attribute probe is enough -- would otherwise be reported as this
// launch's and abort the process at the wrong site (round-5 advisor).
#define DC_LAUNCH(...)                                                                                          \
    do {                                                                                                       \
        (void)hipGetLastError();                                                                               \
        hipLaunchKernelGGL(__VA_ARGS__);                                                                       \
        DC_HIP_CHECK(hipGetLastError());                                                                       \
    } while (0)

typedef unsigned __int128 u128;

// ---- host number theory (setup only; every polynomial operation runs on the GPU) -----------------------
u64 h_mulmod(u64 a, u64 b, u64 q);
u64 h_powmod(u64 a, u64 e, u64 q);
u64 h_invmod(u64 a, u64 q);
bool h_is_prime(u64 n);
// CoeffModulus::Create(2^logN, {bits x count}) ordering: out[0] = last prime found ... out[count-1] = first
bool h_seal_prime_chain(int logN, int bits, int count, std::vector<u64> &out);
u64 h_min_primitive_root(u64 degree, u64 q);
inline u32 h_bitrev(u32 x, int bits)
{
    u32 r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

// prime index of the e-th "other" modulus of digit j at level ell: moduli {0..ell-1} \ {j}, then the special
// prime sp.  e in [0, ell).
__host__ __device__ inline int ks_other_prime(int j, int e, int ell, int sp)
{
    int m = e < j ? e : e + 1;
    return m == ell ? sp : m;
}

// Scratch of one in-flight composite op (key switch / rescale / rotate / mulcc).  Ops running concurrently on
// different HIP streams each own one.
struct Workspace {
    u64 *ks_digits = nullptr; // [Lmax][N]        coefficient-domain digits (also [2][N] staging)
    u64 *ks_ext = nullptr;    // [Lmax][Lmax][N]  digits lifted to every other modulus
    u64 *ks_acc = nullptr;    // [2][Lmax+1][N]   inner products
    u64 *ks_tmp = nullptr;    // [2][Lmax][N]     mod-down correction terms
    u64 *ct_tmp = nullptr;    // [3][Lmax][N]     tensor product / galois scratch
};

struct Context {
    int logN = 0;
    size_t N = 0;
    int K = 0; // primes in the key-level chain; data levels use primes 0..ell-1, special prime = K-1
    // EXTENSION (hybrid_ks.hip; not SEAL's scheme): grouped-digit hybrid key switching -- the last `ksp` primes are special, a digit is
    // a group of `alpha` data primes (alpha <= ksp), keys are [key_digits()][2][K][N].  ksp = alpha = 1 is SEAL's switch_key_inplace.
    int ksp = 1, alpha = 1;
    bool hybrid() const { return ksp > 1 || alpha > 1; }
    int key_digits() const { return hybrid() ? (max_level() + alpha - 1) / alpha : K - 1; }
    int hyb_groups(int ell) const { return (ell + alpha - 1) / alpha; }
    int hyb_ext(int ell) const { return hyb_groups(ell) * (ell + ksp) - ell; } // raised limbs of one key switch at level ell
    int hyb_ext_max() const
    {
        int m = 0;
        for (int l = 1; l <= max_level(); l++) m = hyb_ext(l) > m ? hyb_ext(l) : m;
        return m;
    }
    u64 *d_pmod = nullptr;   // [K]: P mod q_i (P = product of the special primes) for the data primes, 0 for the special ones
    u64 *d_hyb_up = nullptr; // per level ell: qhat_inv[i] (i < ell), then w[i][mi] (mi < ell + ksp) = (Q_g / q_i) mod m_mi
    std::vector<size_t> hyb_up_off;
    int *d_hyb_pidx = nullptr; // per level: prime index of every raised limb, group by group
    std::vector<int> hyb_pidx_off;
    u64 *d_hyb_dn = nullptr; // phat_inv[ksp], half_p[ksp], half_q[L], pinv[L], w_dn[ksp][L]   (L = max_level())
    // matrix-core form of the two base conversions (hybrid_ks.hip, alpha <= 8 and ksp <= 8): the constant matrices as int8 B fragments of
    // v_mfma_i32_16x16x64_i8.  Per level: [group][block of 16 output moduli][8 digit planes][64 lanes][16 bytes]; mod-down: one "group".
    int8_t *d_hyb_bup = nullptr, *d_hyb_bdn = nullptr;
    std::vector<size_t> hyb_bup_off, hyb_bdn_off;
    bool hyb_mfma = false;
    int hyb_up_blocks(int ell) const { return (ell + ksp - 1 + 15) / 16; }  // blocks of 16 "other" moduli (the smallest digit has 1 prime)
    int hyb_dn_blocks(int ell) const { return (ell + 15) / 16; }
    // fused sequence (hybrid_fused.hip): the per-input constants of the two conversions ride on the inverse transforms' last stage -- copies
    // of the per-prime constants whose N^-1 words carry qhat_inv_i (per level: the last digit's composition depends on it) resp. phat_inv_j
    DModulus *d_hyb_upmods = nullptr; // per level ell: [ell]
    std::vector<size_t> hyb_upmods_off;
    DModulus *d_hyb_dnmods = nullptr; // [K]: the special primes' entries modified
    u64 *d_hyb_hp = nullptr;          // [ksp]: floor(P/2) phat_inv_j mod p_j (added after the scaled inverse transform)
    const DModulus *hyb_upmods(int ell) const { return d_hyb_upmods + hyb_upmods_off[(size_t)ell]; }
    const u64 *hyb_up(int ell) const { return d_hyb_up + hyb_up_off[(size_t)ell]; }
    const int *hyb_pidx(int ell) const { return d_hyb_pidx + hyb_pidx_off[(size_t)ell]; }
    int k1 = 0, k2 = 0; // NTT split: COLS phase runs k1 stages, ROWS phase k2 = logN - k1
    std::vector<u64> primes, psi;
    std::vector<DModulus> h_mods;
    DModulus *d_mods = nullptr;
    u64 *d_tw = nullptr;  // [K][N] psi^bitrev(k)
    u64 *d_itw = nullptr; // [K][N] inverse of the above, same index
    // N = 2^15, 60-bit build: the forward table as pairs (w, w 2^31 mod q), 16 bytes per entry, for the single-crossing kernel's twiddle-pair
    // multiply (ntt_full.hip, modarith.hpp mulmod_pair); nullptr otherwise
    u64 *d_tw2 = nullptr;
    // ... and entries 0..1023 of the INVERSE table as pairs + (N^-1, N^-1 2^31) + (N^-1 psi^-bitrev(1), . 2^31): [K][1026][2], the twiddles of the
    // single-crossing kernel's inverse passes B and A (ntt_full.hip full_inv_pass_b_p / full_inv_pass_a_p)
    u64 *d_itw2c = nullptr;
    // 60-bit build, every ring: the first 2^k1 entries of each prime's forward table as pairs, [K][2^k1][2] -- all the twiddles a forward COLS
    // phase uses (they depend on the row group only), for ntt_tile.hpp's pair butterflies; nullptr in the generic-width build.
    // option cols_pairs = 0 makes twc2() return nullptr: the tiles then run on words (A/B measurements, and a second implementation for the tests)
    u64 *d_twc2 = nullptr;
    const u64 *twc2() const { return option(OPT_COLS_PAIRS) ? d_twc2 : nullptr; }
    Workspace ws0;                     // default workspace (kernel-level C ABI, set-up work)
    std::vector<Workspace> workspaces; // everything ever handed out, for the destructor
    u64 *d_inv_last = nullptr;  // [K][K] : inv_last[l*K + i] = q_l^{-1} mod q_i (i != l)
    u64 *d_half_mod = nullptr;  // [K][K] : floor(q_l/2) mod q_i
    int *d_ks_pidx = nullptr;   // per level: prime index of every (digit j, other-modulus e) limb of d_ks_ext
    std::vector<int> ks_pidx_off;
    const int *ks_prime_idx(int ell) const { return d_ks_pidx + ks_pidx_off[ell]; }

    Context(int logN, int K, int bits, const u64 *primes_or_null, int ksp = 1, int alpha = 1);
    ~Context();
    int max_level() const { return K - ksp; }
    void ensure_scratch();
    Workspace new_workspace();
};

} // namespace dacapo
