// Context construction: SEAL-style prime chain + minimal primitive roots + twiddle tables uploaded to HBM.
// Reference behaviour being reproduced: SEAL_HEVM.cpp:46-59 (create_context) / :93-99 (loadSEAL) build a
// SEALContext whose NTT tables use the numerically smallest primitive 2N-th root of each prime, powers
// stored in bit-reversed order [SEAL-upstream numth.cpp, ntt.cpp].
#include "context.hpp"

#include <algorithm>

namespace dacapo {

u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }

u64 h_powmod(u64 a, u64 e, u64 q)
{
    u64 r = 1;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = h_mulmod(r, a, q);
        a = h_mulmod(a, a, q);
    }
    return r;
}

u64 h_invmod(u64 a, u64 q) { return h_powmod(a % q, q - 2, q); }

bool h_is_prime(u64 n)
{
    if (n < 2) return false;
    const u64 small[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
    for (u64 p : small) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    u64 d = n - 1;
    int r = 0;
    while ((d & 1) == 0) d >>= 1, r++;
    for (u64 a : small) { // these 12 witnesses are exact below 2^64
        u64 x = h_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int i = 1; i < r && witness; i++) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

bool h_seal_prime_chain(int logN, int bits, int count, std::vector<u64> &out)
{
    const u64 step = 2ull << logN;
    u64 cand = ((1ull << bits) - 1) / step * step + 1;
    const u64 floor_ = 1ull << (bits - 1);
    std::vector<u64> found;
    for (; (int)found.size() < count && cand > floor_; cand -= step)
        if (h_is_prime(cand)) found.push_back(cand);
    if ((int)found.size() != count) return false;
    out.assign(found.rbegin(), found.rend());
    return true;
}

u64 h_min_primitive_root(u64 degree, u64 q)
{
    if ((q - 1) % degree) return 0;
    u64 g = 0;
    for (u64 x = 2; x < 4096 && !g; x++) {
        u64 c = h_powmod(x, (q - 1) / degree, q);
        if (h_powmod(c, degree >> 1, q) == q - 1) g = c;
    }
    if (!g) return 0;
    const u64 g2 = h_mulmod(g, g, q);
    u64 best = g;
    for (u64 i = 1, cur = g; i < degree / 2; i++) {
        cur = h_mulmod(cur, g2, q);
        if (cur < best) best = cur;
    }
    return best;
}

Context::Context(int logN_, int K_, int bits, const u64 *primes_or_null, int ksp_, int alpha_)
    : logN(logN_), N(1ull << logN_), K(K_), ksp(ksp_), alpha(alpha_)
{
    if (logN < 12 || logN > 17) {
        fprintf(stderr, "[dacapo_amd] ring degree 2^%d unsupported (12..17)\n", logN);
        abort();
    }
    k1 = logN / 2;
    k2 = logN - k1;
    if (primes_or_null)
        primes.assign(primes_or_null, primes_or_null + K);
    else if (!h_seal_prime_chain(logN, bits, K, primes)) {
        fprintf(stderr, "[dacapo_amd] cannot build a %d x %d-bit prime chain for N=2^%d\n", K, bits, logN);
        abort();
    }
    h_mods.resize(K);
    psi.resize(K);
    std::vector<u64> tw((size_t)K * N), itw((size_t)K * N);
    for (int i = 0; i < K; i++) {
        const u64 q = primes[i];
        int qbits = 0;
        while ((q >> qbits) != 0) qbits++;
        const u64 delta = (qbits >= kMinQBits && qbits <= kQBits) ? (1ull << qbits) - q : kMaxDelta;
        if (!DC_GENERIC_WIDTH && qbits != kQBits && delta < kMaxDelta) {
            fprintf(stderr, "[dacapo_amd] prime %d is a %d-bit prime: this build (libSEAL_HEVM.so) is the reference's 60-bit chain only; other widths "
                            "(45..60 bits) run on the generic-width build of the same sources, libSEAL_HEVM_gw.so\n", i, qbits);
            abort();
        }
        if (delta >= kMaxDelta || !prime_shape_ok(q, qbits) || (q - 1) % (2 * N) || !h_is_prime(q)) {
            fprintf(stderr,
                    "[dacapo_amd] prime %d (0x%llx) is not of the form 2^b - d with %d <= b <= %d, d < 2^28, d 2^(64-b) < q, = 1 mod 2N "
                    "(the reference's chain: b = 60, SEAL_HEVM.cpp:48-53)\n", i, (unsigned long long)q, kMinQBits, kQBits);
            abort();
        }
        psi[i] = h_min_primitive_root(2 * N, q);
        const u64 ipsi = h_invmod(psi[i], q);
        u64 pw = 1, ipw = 1;
        u64 *t = tw.data() + (size_t)i * N, *it = itw.data() + (size_t)i * N;
        for (size_t k = 0; k < N; k++) {
            const u32 r = h_bitrev((u32)k, logN);
            t[r] = pw;
            it[r] = ipw;
            pw = h_mulmod(pw, psi[i], q);
            ipw = h_mulmod(ipw, ipsi, q);
        }
        DModulus &m = h_mods[i];
        m.q = q;
        m.delta = fold_word(qbits, delta);
        m.pad_ = 0;
        m.inv_n = h_invmod((u64)N, q);
        m.inv_n_w = h_mulmod(m.inv_n, it[1], q);
    }
    DC_HIP_CHECK(hipMalloc(&d_mods, sizeof(DModulus) * K));
    DC_HIP_CHECK(hipMemcpy(d_mods, h_mods.data(), sizeof(DModulus) * K, hipMemcpyHostToDevice));
    DC_HIP_CHECK(hipMalloc(&d_tw, tw.size() * 8));
    DC_HIP_CHECK(hipMalloc(&d_itw, itw.size() * 8));
    DC_HIP_CHECK(hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    DC_HIP_CHECK(hipMemcpy(d_itw, itw.data(), itw.size() * 8, hipMemcpyHostToDevice));
    if (!DC_GENERIC_WIDTH && logN == 15) { // (the only ring the single-crossing kernel exists for; only its forward passes use pairs)
        std::vector<u64> pr(tw.size() * 2);
        for (int i = 0; i < K; i++)
            for (size_t k = 0; k < N; k++) {
                const u64 w = tw[(size_t)i * N + k];
                pr[2 * ((size_t)i * N + k)] = w, pr[2 * ((size_t)i * N + k) + 1] = h_mulmod(w, 1ull << 31, primes[(size_t)i]);
            }
        DC_HIP_CHECK(hipMalloc(&d_tw2, pr.size() * 8));
        DC_HIP_CHECK(hipMemcpy(d_tw2, pr.data(), pr.size() * 8, hipMemcpyHostToDevice));
        std::vector<u64> ipr((size_t)K * 1026 * 2);
        for (int i = 0; i < K; i++) {
            const u64 q = primes[(size_t)i];
            u64 *row = ipr.data() + (size_t)i * 1026 * 2;
            for (size_t k = 0; k < 1024; k++) row[2 * k] = itw[(size_t)i * N + k], row[2 * k + 1] = h_mulmod(itw[(size_t)i * N + k], 1ull << 31, q);
            row[2 * 1024] = h_mods[(size_t)i].inv_n, row[2 * 1024 + 1] = h_mulmod(h_mods[(size_t)i].inv_n, 1ull << 31, q);
            row[2 * 1025] = h_mods[(size_t)i].inv_n_w, row[2 * 1025 + 1] = h_mulmod(h_mods[(size_t)i].inv_n_w, 1ull << 31, q);
        }
        DC_HIP_CHECK(hipMalloc(&d_itw2c, ipr.size() * 8));
        DC_HIP_CHECK(hipMemcpy(d_itw2c, ipr.data(), ipr.size() * 8, hipMemcpyHostToDevice));
    }

    if (!DC_GENERIC_WIDTH) {
        const size_t n1 = (size_t)1 << k1;
        std::vector<u64> pr((size_t)K * n1 * 2);
        for (int i = 0; i < K; i++)
            for (size_t k = 0; k < n1; k++) {
                const u64 w = tw[(size_t)i * N + k];
                pr[2 * ((size_t)i * n1 + k)] = w, pr[2 * ((size_t)i * n1 + k) + 1] = h_mulmod(w, 1ull << 31, primes[(size_t)i]);
            }
        DC_HIP_CHECK(hipMalloc(&d_twc2, pr.size() * 8));
        DC_HIP_CHECK(hipMemcpy(d_twc2, pr.data(), pr.size() * 8, hipMemcpyHostToDevice));
    }

    // divide-and-round constants for every (dropped prime l, remaining prime i) pair
    std::vector<u64> inv_last((size_t)K * K, 0), half_mod((size_t)K * K, 0);
    for (int l = 0; l < K; l++)
        for (int i = 0; i < K; i++) {
            if (i == l) continue;
            inv_last[(size_t)l * K + i] = h_invmod(primes[l] % primes[i], primes[i]);
            half_mod[(size_t)l * K + i] = (primes[l] >> 1) % primes[i];
        }
    DC_HIP_CHECK(hipMalloc(&d_inv_last, inv_last.size() * 8));
    DC_HIP_CHECK(hipMalloc(&d_half_mod, half_mod.size() * 8));
    DC_HIP_CHECK(hipMemcpy(d_inv_last, inv_last.data(), inv_last.size() * 8, hipMemcpyHostToDevice));
    DC_HIP_CHECK(hipMemcpy(d_half_mod, half_mod.data(), half_mod.size() * 8, hipMemcpyHostToDevice));

    // ---- key-switching constants that depend on the special primes ------------------------------------------------------------
    if (ksp < 1 || alpha < 1 || alpha > ksp || ksp >= K || (hybrid() && (alpha > 16 || ksp > 16 || (max_level() + alpha - 1) / alpha > 16))) {
        fprintf(stderr, "[dacapo_amd] hybrid key switching needs 1 <= alpha <= ksp < K, alpha <= 16 and at most 16 digits (K = %d, ksp = %d, alpha = %d)\n", K,
                ksp, alpha);
        abort();
    }
    const int L = max_level();
    auto prod_except = [&](int lo, int hi, int skip, u64 m) {
        u64 r = 1 % m;
        for (int t = lo; t < hi; t++)
            if (t != skip) r = h_mulmod(r, primes[(size_t)t] % m, m);
        return r;
    };
    std::vector<u64> pmod((size_t)K, 0);
    for (int i = 0; i < L; i++) pmod[(size_t)i] = prod_except(L, K, -1, primes[(size_t)i]);
    DC_HIP_CHECK(hipMalloc(&d_pmod, pmod.size() * 8));
    DC_HIP_CHECK(hipMemcpy(d_pmod, pmod.data(), pmod.size() * 8, hipMemcpyHostToDevice));
    if (hybrid()) {
        std::vector<u64> up;
        std::vector<int> pidx;
        hyb_up_off.assign((size_t)L + 2, 0), hyb_pidx_off.assign((size_t)L + 2, 0);
        for (int ell = 1; ell <= L; ell++) {
            const int M = ell + ksp;
            hyb_up_off[(size_t)ell] = up.size(), hyb_pidx_off[(size_t)ell] = (int)pidx.size();
            const size_t base = up.size();
            up.resize(base + (size_t)ell + (size_t)ell * M, 0);
            for (int g = 0; g < hyb_groups(ell); g++) {
                const int lo = g * alpha, hi = std::min(lo + alpha, ell);
                for (int i = lo; i < hi; i++) {
                    up[base + (size_t)i] = h_invmod(prod_except(lo, hi, i, primes[(size_t)i]), primes[(size_t)i]);
                    for (int mi = 0; mi < M; mi++) {
                        const int pm = mi < ell ? mi : L + (mi - ell);
                        up[base + (size_t)ell + (size_t)i * M + mi] = prod_except(lo, hi, i, primes[(size_t)pm]);
                    }
                }
                for (int mi = 0; mi < M; mi++)
                    if (mi < lo || mi >= hi) pidx.push_back(mi < ell ? mi : L + (mi - ell));
            }
        }
        DC_HIP_CHECK(hipMalloc(&d_hyb_up, up.size() * 8));
        DC_HIP_CHECK(hipMemcpy(d_hyb_up, up.data(), up.size() * 8, hipMemcpyHostToDevice));
        DC_HIP_CHECK(hipMalloc(&d_hyb_pidx, pidx.size() * sizeof(int)));
        DC_HIP_CHECK(hipMemcpy(d_hyb_pidx, pidx.data(), pidx.size() * sizeof(int), hipMemcpyHostToDevice));
        std::vector<u64> dn((size_t)2 * ksp + 2 * L + (size_t)ksp * L, 0);
        for (int j = 0; j < ksp; j++) {
            const u64 pj = primes[(size_t)(L + j)];
            dn[(size_t)j] = h_invmod(prod_except(L, K, L + j, pj), pj);
            dn[(size_t)(ksp + j)] = (pj - 1) >> 1; // floor(P/2) mod p_j: P = 0 mod p_j and odd
            for (int i = 0; i < L; i++) dn[(size_t)2 * ksp + 2 * L + (size_t)j * L + i] = prod_except(L, K, L + j, primes[(size_t)i]);
        }
        for (int i = 0; i < L; i++) {
            const u64 qi = primes[(size_t)i], Pq = pmod[(size_t)i];
            dn[(size_t)(2 * ksp + i)] = h_mulmod((Pq + qi - 1) % qi, (qi + 1) >> 1, qi); // floor(P/2) = (P - 1) / 2 mod q_i
            dn[(size_t)(2 * ksp + L + i)] = h_invmod(Pq, qi);
        }
        DC_HIP_CHECK(hipMalloc(&d_hyb_dn, dn.size() * 8));
        DC_HIP_CHECK(hipMemcpy(d_hyb_dn, dn.data(), dn.size() * 8, hipMemcpyHostToDevice));
        { // the conversions' per-input constants folded into the inverse transforms' N^-1 words (hybrid_fused.hip)
            std::vector<DModulus> um;
            hyb_upmods_off.assign((size_t)L + 2, 0);
            for (int ell = 1; ell <= L; ell++) {
                hyb_upmods_off[(size_t)ell] = um.size();
                const u64 *upl = up.data() + hyb_up_off[(size_t)ell];
                for (int i = 0; i < ell; i++) {
                    DModulus m = h_mods[(size_t)i];
                    m.inv_n = h_mulmod(m.inv_n, upl[i], m.q), m.inv_n_w = h_mulmod(m.inv_n_w, upl[i], m.q);
                    um.push_back(m);
                }
            }
            DC_HIP_CHECK(hipMalloc(&d_hyb_upmods, um.size() * sizeof(DModulus)));
            DC_HIP_CHECK(hipMemcpy(d_hyb_upmods, um.data(), um.size() * sizeof(DModulus), hipMemcpyHostToDevice));
            std::vector<DModulus> dm(h_mods.begin(), h_mods.end());
            std::vector<u64> hp((size_t)ksp);
            for (int j = 0; j < ksp; j++) {
                DModulus &m = dm[(size_t)(L + j)];
                m.inv_n = h_mulmod(m.inv_n, dn[(size_t)j], m.q), m.inv_n_w = h_mulmod(m.inv_n_w, dn[(size_t)j], m.q);
                hp[(size_t)j] = h_mulmod(dn[(size_t)(ksp + j)], dn[(size_t)j], m.q);
            }
            DC_HIP_CHECK(hipMalloc(&d_hyb_dnmods, dm.size() * sizeof(DModulus)));
            DC_HIP_CHECK(hipMemcpy(d_hyb_dnmods, dm.data(), dm.size() * sizeof(DModulus), hipMemcpyHostToDevice));
            DC_HIP_CHECK(hipMalloc(&d_hyb_hp, hp.size() * 8));
            DC_HIP_CHECK(hipMemcpy(d_hyb_hp, hp.data(), hp.size() * 8, hipMemcpyHostToDevice));
        }
        // ---- the same two constant matrices as int8 operands of the matrix cores -------------------------------------------------
        // A residue y < 2^60 is 8 balanced base-256 digits (y + 0x80..80 with every byte's top bit flipped: digits in [-128, 127]); a
        // matrix entry w enters as the balanced digits of V_p = w 2^(8p) mod m for each of the operand's 8 digit positions p, so that
        //     sum_t y_t w_t  =  sum_r 2^(8r) [ sum_{t,p} digit_p(y_t) digit_r(V_{t,p}) ]   (mod m)
        // and the bracket is one int8 dot product of length 8 |S| <= 64 = ONE v_mfma_i32_16x16x64_i8 per (16 coefficients, 16 moduli, r).
        // B fragment of lane l: column l & 15, k = 16 (l >> 4) + j in byte j, k = 8 t + p.
        // (round 5: up to 12 special primes -- the mod-down's inputs 8..11 go through a second, 32-deep MFMA; their B fragments, 8 bytes per
        // lane, follow the level's main table: [block][plane][lane])
        hyb_mfma = alpha <= 8 && ksp <= 12 && option(OPT_HYB_MFMA) != 0;
        if (hyb_mfma) {
            auto balanced = [](u64 v, int8_t *out8) {
                const u64 C = 0x8080808080808080ull, b = (v + C) ^ C;
                for (int r = 0; r < 8; r++) out8[r] = (int8_t)(uint8_t)(b >> (8 * r));
            };
            std::vector<int8_t> bup, bdn;
            hyb_bup_off.assign((size_t)L + 2, 0), hyb_bdn_off.assign((size_t)L + 2, 0);
            for (int ell = 1; ell <= L; ell++) {
                const int M = ell + ksp, nb = hyb_up_blocks(ell), G = hyb_groups(ell);
                hyb_bup_off[(size_t)ell] = bup.size();
                const size_t base = bup.size();
                bup.resize(base + (size_t)G * nb * 8 * 64 * 16, 0);
                const u64 *upl = up.data() + hyb_up_off[(size_t)ell];
                for (int g = 0; g < G; g++) {
                    const int lo = g * alpha, hi = std::min(lo + alpha, ell), a = hi - lo;
                    for (int e = 0; e < M - a; e++) {
                        const int mi = e < lo ? e : e + a, pm = mi < ell ? mi : L + (mi - ell), blk = e / 16, col = e % 16;
                        const u64 m = primes[(size_t)pm];
                        for (int t = 0; t < a; t++) {
                            u64 V = upl[(size_t)ell + (size_t)(lo + t) * M + mi] % m;
                            for (int pp = 0; pp < 8; pp++) {
                                int8_t d8[8];
                                balanced(V, d8);
                                const int k = 8 * t + pp, lane = (k / 16) * 16 + col, j = k % 16;
                                for (int r = 0; r < 8; r++)
                                    bup[base + ((((size_t)g * nb + blk) * 8 + r) * 64 + lane) * 16 + j] = d8[r];
                                V = h_mulmod(V, 256 % m, m);
                            }
                        }
                    }
                }
                // mod-down: inputs z_j (j < ksp), outputs q_i (i < ell)
                const int nd = hyb_dn_blocks(ell);
                hyb_bdn_off[(size_t)ell] = bdn.size();
                const size_t bd = bdn.size(), tail = bd + (size_t)nd * 8 * 64 * 16;
                bdn.resize(tail + (ksp > 8 ? (size_t)nd * 8 * 64 * 8 : 0), 0);
                for (int i = 0; i < ell; i++) {
                    const u64 m = primes[(size_t)i];
                    for (int j2 = 0; j2 < ksp; j2++) {
                        u64 V = dn[(size_t)2 * ksp + 2 * L + (size_t)j2 * L + i] % m;
                        for (int pp = 0; pp < 8; pp++) {
                            int8_t d8[8];
                            balanced(V, d8);
                            if (j2 < 8) { // main chunk, v_mfma_i32_16x16x64_i8: lane = 16 (k / 16) + column, byte k % 16, k = 8 j + p
                                const int k = 8 * j2 + pp, lane = (k / 16) * 16 + (i % 16), j = k % 16;
                                for (int r = 0; r < 8; r++) bdn[bd + ((((size_t)(i / 16)) * 8 + r) * 64 + lane) * 16 + j] = d8[r];
                            } else {      // tail, v_mfma_i32_16x16x32_i8: lane = 16 (k / 8) + column, byte k % 8, k = 8 (j - 8) + p
                                const int lane = (j2 - 8) * 16 + (i % 16);
                                for (int r = 0; r < 8; r++) bdn[tail + ((((size_t)(i / 16)) * 8 + r) * 64 + lane) * 8 + pp] = d8[r];
                            }
                            V = h_mulmod(V, 256 % m, m);
                        }
                    }
                }
            }
            DC_HIP_CHECK(hipMalloc(&d_hyb_bup, bup.size()));
            DC_HIP_CHECK(hipMemcpy(d_hyb_bup, bup.data(), bup.size(), hipMemcpyHostToDevice));
            DC_HIP_CHECK(hipMalloc(&d_hyb_bdn, bdn.size()));
            DC_HIP_CHECK(hipMemcpy(d_hyb_bdn, bdn.data(), bdn.size(), hipMemcpyHostToDevice));
        }
    }
}

Workspace Context::new_workspace()
{
    const size_t L = max_level();
    Workspace w;
    DC_HIP_CHECK(hipMalloc(&w.ks_digits, std::max<size_t>(L, 2) * N * 8));
    const size_t ext = std::max<size_t>(std::max<size_t>(L * L, 3 * (size_t)K), hybrid() ? (size_t)hyb_ext_max() : 0);
    DC_HIP_CHECK(hipMalloc(&w.ks_ext, ext * N * 8)); // also [3][K][N] encryption staging
    DC_HIP_CHECK(hipMalloc(&w.ks_acc, 2 * (L + (size_t)ksp) * N * 8));
    DC_HIP_CHECK(hipMalloc(&w.ks_tmp, 2 * std::max<size_t>(L, 1) * N * 8));
    DC_HIP_CHECK(hipMalloc(&w.ct_tmp, 3 * std::max<size_t>(L, 1) * N * 8));
    workspaces.push_back(w);
    return w;
}

void Context::ensure_scratch()
{
    if (ws0.ks_digits) return;
    const size_t L = max_level();
    ws0 = new_workspace();
    std::vector<int> pidx;
    ks_pidx_off.assign(L + 2, 0);
    for (int ell = 1; ell <= (int)L; ell++) {
        ks_pidx_off[ell] = (int)pidx.size();
        for (int j = 0; j < ell; j++)
            for (int e = 0; e < ell; e++) pidx.push_back(ks_other_prime(j, e, ell, K - ksp));
    }
    DC_HIP_CHECK(hipMalloc(&d_ks_pidx, std::max<size_t>(pidx.size(), 1) * sizeof(int)));
    DC_HIP_CHECK(hipMemcpy(d_ks_pidx, pidx.data(), pidx.size() * sizeof(int), hipMemcpyHostToDevice));
}

Context::~Context()
{
    for (void *p : { (void *)d_mods, (void *)d_tw, (void *)d_itw, (void *)d_tw2, (void *)d_itw2c, (void *)d_twc2, (void *)d_inv_last, (void *)d_half_mod, (void *)d_ks_pidx, (void *)d_pmod,
                     (void *)d_hyb_up, (void *)d_hyb_pidx, (void *)d_hyb_dn, (void *)d_hyb_bup, (void *)d_hyb_bdn, (void *)d_hyb_upmods, (void *)d_hyb_dnmods,
                     (void *)d_hyb_hp })
        if (p) (void)hipFree(p);
    for (Workspace &w : workspaces)
        for (void *p : { (void *)w.ks_digits, (void *)w.ks_ext, (void *)w.ks_acc, (void *)w.ks_tmp, (void *)w.ct_tmp })
            if (p) (void)hipFree(p);
}

} // namespace dacapo
