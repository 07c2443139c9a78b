// Plan construction and execution for the HEVM VM (see plan.hpp for the idea).  Semantics per opcode are those of
// SEAL_HEVM.cpp:268-334 (including the scale overwrite of addcc/addcp at :301,:308 and the untouched dst of a
// zero modswitch at :288), evaluated once at plan time in program order.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <functional>
#include <queue>
#include <set>
#include <tuple>

#include "hevm_vm.hpp"

namespace dacapo {

#define ks_ntts(ell) ks_ntt_count(c, (ell))

template <class T>
static T *upload(const std::vector<T> &v)
{
    T *d = nullptr;
    DC_HIP_CHECK(vm_malloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) DC_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

void HEVM::build_plan()
{
    Context &c = *ctx;
    const size_t N = c.N;
    Plan &P = plan;
    for (u64 *b : P.handoff_bufs) (void)vm_free(b);
    P.handoff_bufs.clear();
    for (void *p : { (void *)P.d_enc_items, (void *)P.enc_arena, (void *)P.enc_scratch, (void *)P.d_enc_overflow })
        if (p) (void)vm_free(p);
    P.d_enc_items = nullptr, P.enc_arena = nullptr, P.enc_scratch = nullptr, P.d_enc_overflow = nullptr;
    P.enc_groups.clear(), P.enc_arena_bytes = P.enc_scratch_bytes = 0;
    if (P.d_cont_other) (void)vm_free(P.d_cont_other), P.d_cont_other = nullptr;
    for (void *p : { (void *)P.d_ks, (void *)P.d_mul, (void *)P.d_rs, (void *)P.d_ew, (void *)P.d_sum, (void *)P.d_sum_srcs, (void *)P.d_sumg, (void *)P.d_sumg_srcs, (void *)P.d_boot,
                     (void *)P.d_boot_rs, (void *)P.zenc, (void *)P.boot_ue, (void *)P.boot_tmp, (void *)P.boot_pt[0], (void *)P.boot_ptx[0],
                     (void *)P.boot_pt[1], (void *)P.boot_ptx[1] })
        if (p) (void)vm_free(p);
    P.d_boot = nullptr, P.d_boot_rs = nullptr, P.zenc = P.boot_ue = P.boot_tmp = nullptr;
    P.boot_pt[0] = P.boot_pt[1] = P.boot_ptx[0] = P.boot_ptx[1] = nullptr;
    P.boot_chunks.clear();
    if (P.graph_exec) (void)hipGraphExecDestroy(P.graph_exec);
    if (P.graph) (void)hipGraphDestroy(P.graph);
    P.graph_exec = nullptr, P.graph = nullptr;
    P.vals.clear(), P.pops.clear(), P.steps.clear();
    P.n_keyswitch = P.n_ntt = 0;
    const size_t nreg = ciphers.size();
    const int S = streams; // independent ciphertext streams executed side by side (items of every step are replicated)
    std::vector<int> cur(nreg, -1);
    auto new_val = [&](int level, double scale) {
        Val v;
        v.level = level, v.scale = scale, v.root = (int)P.vals.size();
        P.vals.push_back(v);
        return (int)P.vals.size() - 1;
    };
    auto need = [&](int r, const char *what) {
        if (r < 0 || (size_t)r >= nreg || cur[(size_t)r] < 0) {
            fprintf(stderr, "[dacapo_amd] %s reads cipher register %d before anything wrote it\n", what, r);
            abort();
        }
        return cur[(size_t)r];
    };
    auto add_pop = [&](PopKind k, int level, std::vector<int> srcs, int dst) -> Pop & {
        Pop p;
        p.kind = k, p.level = level, p.srcs = std::move(srcs), p.dst = dst;
        P.vals[(size_t)dst].def_pop = (int)P.pops.size();
        P.pops.push_back(p);
        return P.pops.back();
    };
    for (size_t i = 0; i < header.arg_length; i++) { // program inputs: written by encrypt() into the home buffers
        const int v = new_val((int)arg_level[i], pow(2.0, (double)arg_scale[i]));
        P.vals[(size_t)v].external = true;
        P.vals[(size_t)v].buf = home.at(i);
        cur[i] = v;
    }
    // ---- 1. SSA walk in program order: metadata semantics of the reference, one pseudo-op per kernel sequence ----
    // option rot_compose: (value, Galois element) -> the value that hop produced.  A memoised value is only ever used as an INTERMEDIATE
    // or as the buffer behind a register's own view: when a rotate's last hop hits the memo, dst gets a fresh value sharing the buffer
    // (as modswitch / setscale views do), so that the reference's scale overwrites (addcc / addcp touch ONE register's metadata,
    // SEAL_HEVM.cpp:301,308) never reach another register that names the same limbs, and a rotation's result carries the scale its
    // operand has NOW, not the one the memoised hop saw.  memo_base maps such a view back to the value whose hops are memoised.
    std::map<std::pair<int, u32>, int> hop_memo;
    std::map<int, int> memo_base;
    for (const WireOp &op : ops) {
        switch (op.opcode) {
        case 1: { // rotate: one key-switch hop per Galois element (direct key or NAF digits)
            int v = need(op.lhs, "rotate");
            const double in_scale = P.vals[(size_t)v].scale; // (rotate_vector keeps the scale: every hop's result carries the operand's)
            bool memo_final = false;
            const std::vector<u32> hops = rotate_hops((int16_t)op.rhs);
            for (const u32 &elt : hops) {
                // option rot_compose (a bounded key set): composed rotations of one value mostly start with the same small offset -- the parts
                // come in ascending order -- so the hop (value, Galois element) is computed once and its result named again: the same limbs
                // (a key switch is deterministic), one key switch fewer.  config 4 under the reference HEaaN runtime's 49 keys: 18.7 % of the hops.
                memo_final = false;
                if (rot_compose) {
                    const auto mb = memo_base.find(v);
                    auto it = hop_memo.find({ mb != memo_base.end() ? mb->second : v, elt });
                    if (it != hop_memo.end()) {
                        v = it->second;
                        memo_final = true;
                        continue;
                    }
                }
                const Val s = P.vals[(size_t)v];
                const int nv = new_val(s.level, in_scale);
                Pop &p = add_pop(P_ROT, s.level, { v }, nv);
                p.elt = elt, p.key = keys.galois.at(elt);
                p.op = (int)(&op - ops.data()), p.direct = &elt == &hops.back(); // (the LAST hop of the instruction: the one whose result the program sees)
                P.n_keyswitch++, P.n_ntt += ks_ntts(s.level);
                if (rot_compose) {
                    const auto mb = memo_base.find(v);
                    hop_memo[{ mb != memo_base.end() ? mb->second : v, elt }] = nv;
                }
                v = nv;
            }
            if (memo_final) { // the last hop came out of the memo: another register may name that value -- dst gets its own view of the buffer
                const Val s = P.vals[(size_t)v];
                const int nv = new_val(s.level, in_scale);
                P.vals[(size_t)nv].root = s.root;
                P.vals[(size_t)nv].def_pop = P.vals[(size_t)s.root].def_pop;
                P.vals[(size_t)v].uses++; // the view keeps the memoised value alive and observable
                const auto mb = memo_base.find(v);
                memo_base[nv] = mb != memo_base.end() ? mb->second : v;
                v = nv;
            }
            if (v == cur[op.lhs] && op.dst != op.lhs) { // no hop at all: rotate_vector copies (SEAL_HEVM.cpp:273), so dst gets a value of its
                const Val s = P.vals[(size_t)v];            // own -- a later scale overwrite on one register (:301,:308) must not reach the other
                const int nv = new_val(s.level, s.scale);
                add_pop(P_COPY, s.level, { v }, nv);
                v = nv;
            }
            cur[op.dst] = v;
            break;
        }
        case 2: {
            const int a = need(op.lhs, "negate");
            const Val s = P.vals[(size_t)a];
            const int nv = new_val(s.level, s.scale);
            add_pop(P_NEG, s.level, { a }, nv);
            cur[op.dst] = nv;
            break;
        }
        case 3: {
            const int a = need(op.lhs, "rescale");
            const Val s = P.vals[(size_t)a];
            if (s.level < 2) {
                fprintf(stderr, "[dacapo_amd] rescale: end of modulus switching chain reached\n");
                abort();
            }
            const int nv = new_val(s.level - 1, s.scale / (double)c.primes[(size_t)s.level - 1]);
            add_pop(P_RESCALE, s.level, { a }, nv);
            P.n_ntt += 2 * s.level;
            cur[op.dst] = nv;
            break;
        }
        case 4: { // CKKS mod_switch_to_next drops limbs: a view of the same buffer with fewer primes
            const int down = (int16_t)op.rhs;
            if (down <= 0) break; // dst untouched (SEAL_HEVM.cpp:288)
            const int a = need(op.lhs, "modswitch");
            const Val s = P.vals[(size_t)a];
            if (s.level - down < 1) {
                fprintf(stderr, "[dacapo_amd] modswitch: end of modulus switching chain reached\n");
                abort();
            }
            const int nv = new_val(s.level - down, s.scale);
            P.vals[(size_t)nv].root = s.root;
            P.vals[(size_t)nv].def_pop = P.vals[(size_t)s.root].def_pop;
            P.vals[(size_t)a].uses++; // the view keeps the original alive and observable
            cur[op.dst] = nv;
            break;
        }
        case 5: fprintf(stderr, "This VM does not support native upscale op\n"); abort();
        case 6: {
            const int a = need(op.lhs, "addcc"), b = need(op.rhs, "addcc");
            P.vals[(size_t)a].scale = P.vals[(size_t)b].scale; // SEAL_HEVM.cpp:301
            if (P.vals[(size_t)a].level != P.vals[(size_t)b].level) {
                fprintf(stderr, "[dacapo_amd] addcc: level mismatch %d vs %d\n", P.vals[(size_t)a].level, P.vals[(size_t)b].level);
                abort();
            }
            const int nv = new_val(P.vals[(size_t)a].level, P.vals[(size_t)b].scale);
            add_pop(P_SUM, P.vals[(size_t)a].level, { a, b }, nv);
            cur[op.dst] = nv;
            break;
        }
        case 7: {
            const int a = need(op.lhs, "addcp");
            const Plain &pl = plains.at(op.rhs);
            P.vals[(size_t)a].scale = pl.scale; // SEAL_HEVM.cpp:308
            if (P.vals[(size_t)a].level != pl.level) {
                fprintf(stderr, "[dacapo_amd] addcp: level mismatch %d vs %d\n", P.vals[(size_t)a].level, pl.level);
                abort();
            }
            const int nv = new_val(pl.level, pl.scale);
            add_pop(P_ADDP, pl.level, { a }, nv).plain = op.rhs;
            cur[op.dst] = nv;
            break;
        }
        case 8: {
            const int a = need(op.lhs, "mulcc"), b = need(op.rhs, "mulcc");
            const Val sa = P.vals[(size_t)a], sb = P.vals[(size_t)b];
            if (sa.level != sb.level) {
                fprintf(stderr, "[dacapo_amd] mulcc: level mismatch %d vs %d\n", sa.level, sb.level);
                abort();
            }
            const int nv = new_val(sa.level, sa.scale * sb.scale);
            add_pop(P_MULCC, sa.level, { a, b }, nv);
            P.n_keyswitch++, P.n_ntt += ks_ntts(sa.level);
            cur[op.dst] = nv;
            break;
        }
        case 9: {
            const int a = need(op.lhs, "mulcp");
            const Val sa = P.vals[(size_t)a];
            const Plain &pl = plains.at(op.rhs);
            if (sa.level != pl.level) {
                fprintf(stderr, "[dacapo_amd] mulcp: level mismatch %d vs %d\n", sa.level, pl.level);
                abort();
            }
            const int nv = new_val(sa.level, sa.scale * pl.scale);
            add_pop(P_MULP, sa.level, { a }, nv).plain = op.rhs;
            cur[op.dst] = nv;
            break;
        }
        case 10: {
            const int a = need(op.lhs, "bootstrap");
            const Val sa = P.vals[(size_t)a];
            const int nv = new_val((int)op.rhs, pow(2.0, (double)(int64_t)std::log2(sa.scale))); // SEAL_HEVM.cpp:332
            add_pop(P_BOOT, sa.level, { a }, nv).target_level = op.rhs;
            (void)crt_tables(sa.level); // device CRT constants are uploaded now, never during a (captured) run
            cur[op.dst] = nv;
            break;
        }
        case kOpConj: { // one key switch with the conjugation key
            const int a = need(op.lhs, "conj");
            const Val s = P.vals[(size_t)a];
            const u32 elt = (u32)(2 * N - 1);
            if (!keys.galois.count(elt)) {
                fprintf(stderr, "[dacapo_amd] conj: no Galois key for the conjugation\n");
                abort();
            }
            const int nv = new_val(s.level, s.scale);
            Pop &p = add_pop(P_ROT, s.level, { a }, nv);
            p.elt = elt, p.key = keys.galois.at(elt);
            P.n_keyswitch++, P.n_ntt += ks_ntts(s.level);
            cur[op.dst] = nv;
            break;
        }
        case kOpModRaise: {
            const int a = need(op.lhs, "modraise");
            const Val s = P.vals[(size_t)a];
            if (s.level != 1 || op.rhs < 1 || (int)op.rhs > c.max_level()) {
                fprintf(stderr, "[dacapo_amd] modraise: the operand must sit at 1 prime (has %d) and the target within 1..%d\n", s.level, c.max_level());
                abort();
            }
            const int nv = new_val((int)op.rhs, s.scale);
            add_pop(P_MODRAISE, 1, { a }, nv).target_level = op.rhs;
            P.n_ntt += 2 + 2 * (int)op.rhs;
            cur[op.dst] = nv;
            break;
        }
        case kOpSetScale: { // a relabelled view of the same buffer
            const int a = need(op.lhs, "setscale");
            const Val s = P.vals[(size_t)a];
            const int nv = new_val(s.level, buffer[op.rhs][0]); // validated by load_program
            P.vals[(size_t)nv].root = s.root;
            P.vals[(size_t)nv].def_pop = P.vals[(size_t)s.root].def_pop;
            P.vals[(size_t)a].uses++;
            cur[op.dst] = nv;
            break;
        }
        default: break;
        }
    }
    P.final_val = cur;
    std::vector<Val> &V = P.vals;
    std::vector<Pop> &O = P.pops;
    for (const Pop &p : O)
        for (int s : p.srcs) V[(size_t)s].uses++;
    for (int v : cur)
        if (v >= 0) V[(size_t)v].uses++, V[(size_t)V[(size_t)v].root].pinned = true;

    // ---- 2. fold ct+ct chains into n-ary sums (only through intermediates nothing else observes) ------------------
    for (Pop &p : O) {
        if (p.kind != P_SUM) continue;
        std::vector<int> flat;
        for (int s : p.srcs) {
            const Val &sv = V[(size_t)s];
            const int dp = sv.root == s ? sv.def_pop : -1;
            if (dp >= 0 && O[(size_t)dp].kind == P_SUM && !O[(size_t)dp].dead && sv.uses == 1 && O[(size_t)dp].dst == s) {
                flat.insert(flat.end(), O[(size_t)dp].srcs.begin(), O[(size_t)dp].srcs.end());
                O[(size_t)dp].dead = true;
            } else
                flat.push_back(s);
        }
        p.srcs = flat;
    }
    // ... and fold single-use ct*pt products into the sum that consumes them (one pass over a, pt instead of
    // write product / read product)
    for (Pop &p : O) {
        if (p.kind != P_SUM || p.dead) continue;
        p.src_plain.assign(p.srcs.size(), -1);
        for (size_t k = 0; k < p.srcs.size(); k++) {
            const int sidx = p.srcs[k];
            const Val &sv = V[(size_t)sidx];
            const int dp = sv.root == sidx ? sv.def_pop : -1;
            if (dp >= 0 && O[(size_t)dp].kind == P_MULP && !O[(size_t)dp].dead && sv.uses == 1 && O[(size_t)dp].dst == sidx) {
                p.srcs[k] = O[(size_t)dp].srcs[0];
                p.src_plain[k] = O[(size_t)dp].plain;
                O[(size_t)dp].dead = true;
            }
        }
    }

    // ... and fold the elementwise producers of a rescale's operand into the rescale itself: the operand is read exactly
    // twice (dropped limb by the first inverse phase, the other limbs by the last kernel), so a single-use
    // mulcp / addcp / n-ary sum in front of it need not be materialised:  rescale(((sum) + a) * m)
    for (Pop &p : O) {
        if (p.kind != P_RESCALE || p.dead) continue;
        auto single_def = [&](int v, PopKind k) -> int { // pop defining v if v is a plain (non-view) single-use value made by kind k
            const Val &sv = V[(size_t)v];
            const int dp = sv.root == v ? sv.def_pop : -1;
            if (dp >= 0 && O[(size_t)dp].kind == k && !O[(size_t)dp].dead && sv.uses == 1 && O[(size_t)dp].dst == v && !sv.pinned) return dp;
            return -1;
        };
        int x = p.srcs[0], dp;
        if ((dp = single_def(x, P_MULP)) >= 0) p.rs_mul = O[(size_t)dp].plain, x = O[(size_t)dp].srcs[0], O[(size_t)dp].dead = true;
        if ((dp = single_def(x, P_ADDP)) >= 0) p.rs_add = O[(size_t)dp].plain, x = O[(size_t)dp].srcs[0], O[(size_t)dp].dead = true;
        p.srcs[0] = x;
        if ((dp = single_def(x, P_SUM)) >= 0) {
            p.rs_sum = true;
            p.srcs = O[(size_t)dp].srcs, p.src_plain = O[(size_t)dp].src_plain;
            O[(size_t)dp].dead = true;
        }
    }

    // ... and a rescale whose only consumer is an opcode 10 is not executed at all.  Opcode 10 decrypts, re-encodes and
    // re-encrypts with fresh randomness; decrypt(rescale(x)) = (decrypt(x) - (d0 + d1*s)) / q with |d| <= q/2 the rounding
    // remainders, i.e. decrypt(x) / q up to a term of a few dozen units in a 2^40-scaled coefficient (1e-11 relative, far
    // under the ciphertext's own noise), so the re-encoder reads x itself at one prime more and divides by q in its
    // double-precision step.  (Limb-level equality with the reference never extends past an opcode 10 anyway.)
    for (Pop &p : O) {
        if (p.kind != P_BOOT || p.dead || !fold_rescale_into_boot) continue;
        const int x = p.srcs[0];
        const Val &sv = V[(size_t)x];
        const int dp = sv.root == x ? sv.def_pop : -1;
        if (dp < 0 || O[(size_t)dp].kind != P_RESCALE || O[(size_t)dp].dead || sv.uses != 1 || O[(size_t)dp].dst != x || sv.pinned) continue;
        Pop &r = O[(size_t)dp];
        p.srcs = r.srcs, p.src_plain = r.src_plain, p.rs_sum = r.rs_sum, p.rs_add = r.rs_add, p.rs_mul = r.rs_mul;
        p.level = r.level, p.boot_drop = r.level - 1, p.boot_src_scale = sv.scale;
        (void)crt_tables(p.level);
        P.n_ntt -= 2 * r.level;
        r.dead = true;
    }

    // ---- 2b. lazy sums (option hyb_lazy_sum; grouped-digit mode) ------------------------------------------------------------------
    // Rotations whose results are only ever added together -- the giant steps of a BSGS matrix-vector product: out = sum_g rot_g(inner_g) --
    // share ONE division by P ("double hoisting", Bossuat et al. 2021): the accumulators of the group's key switches are added in the raised
    // basis and F6 ... F9 run once (hybrid_fused.hip hybf_rotate_sum), 1 instead of n roundings.  That is a different -- slightly less noisy --
    // result than n rotate instructions, limb-wise, which is why the option is off by default and why the rule is narrow and exported
    // (hevm_plan_lazy_groups; the oracle VM replays exactly these groups): a member is the LAST hop of a rotate instruction (the only one under a
    // direct key) whose result nothing else reads, entering the sum as it is (no plaintext factor); a sum needs at least two of them.
    if (lazy_sums && hyb_lazy_sum_supported(c)) {
        for (size_t ci = 0; ci < O.size(); ci++) {
            if (O[ci].dead) continue;
            if (!(O[ci].kind == P_SUM || ((O[ci].kind == P_RESCALE || O[ci].kind == P_BOOT) && O[ci].rs_sum))) continue;
            std::vector<size_t> terms; // positions in the consumer's source list
            for (size_t k = 0; k < O[ci].srcs.size(); k++) {
                // a term times a plaintext joins only under option hyb_double_hoist: its product is then taken in the raised basis, which needs
                // the plaintext over the special primes too (encoded below, from the resident constants)
                if (!O[ci].src_plain.empty() && O[ci].src_plain[k] >= 0 &&
                    !(double_hoist && !online_encode && !host_encoder && dh_items.count(O[ci].src_plain[k]) && plains.at((size_t)O[ci].src_plain[k]).level >= O[ci].level))
                    continue;
                const int sidx = O[ci].srcs[k];
                const Val &sv = V[(size_t)sidx];
                const int dp = sv.root == sidx ? sv.def_pop : -1;
                if (dp < 0) continue;
                const Pop &r = O[(size_t)dp];
                if (r.kind == P_ROT && !r.dead && r.direct && r.dst == sidx && sv.uses == 1 && !sv.pinned && r.level == O[ci].level) terms.push_back(k);
            }
            if (terms.size() < 2) continue;
            // the group takes the place of its LAST member in pop order: every member's source is defined before it, its consumer after it
            std::sort(terms.begin(), terms.end(), [&](size_t x, size_t y) { return V[(size_t)O[ci].srcs[x]].def_pop < V[(size_t)O[ci].srcs[y]].def_pop; });
            const size_t keep = terms.back();
            Pop grp = O[(size_t)V[(size_t)O[ci].srcs[keep]].def_pop];
            grp.kind = P_ROTSUM, grp.srcs.clear(), grp.direct = false, grp.op = -1;
            for (size_t k : terms) {
                Pop &r = O[(size_t)V[(size_t)O[ci].srcs[k]].def_pop];
                grp.srcs.push_back(r.srcs[0]), grp.elts.push_back(r.elt), grp.keys.push_back(r.key), grp.ops.push_back(r.op);
                grp.plains.push_back(O[ci].src_plain.empty() ? -1 : O[ci].src_plain[k]);
                r.dead = true;
                if (k != keep) V[(size_t)O[ci].srcs[k]].uses = 0, V[(size_t)O[ci].srcs[k]].def_pop = -1; // (no pop defines or reads this value any more)
            }
            const int gi = V[(size_t)O[ci].srcs[keep]].def_pop;
            O[(size_t)gi] = grp;
            std::vector<int> srcs, plain;
            for (size_t k = 0; k < O[ci].srcs.size(); k++)
                if (k == keep || std::find(terms.begin(), terms.end(), k) == terms.end()) {
                    srcs.push_back(O[ci].srcs[k]);
                    if (!O[ci].src_plain.empty()) plain.push_back(k == keep ? -1 : O[ci].src_plain[k]); // (the group applies its members' plaintexts itself)
                }
            O[ci].srcs = srcs, O[ci].src_plain = plain;
            // A plain sum whose terms were ALL grouped (rot(x, a) + rot(y, b)) is now a one-source sum without a plaintext: a copy of the group's
            // result.  The group writes the sum's destination itself and the sum goes away (round-5 advisor: one launch and one buffer less).
            if (O[ci].kind == P_SUM && O[ci].srcs.size() == 1 && (O[ci].src_plain.empty() || O[ci].src_plain[0] < 0)) {
                const int old = O[(size_t)gi].dst, dst = O[ci].dst;
                if (V[(size_t)dst].root == dst && V[(size_t)dst].level == V[(size_t)old].level) { // (its own buffer, same number of limbs)
                    O[(size_t)gi].dst = dst;
                    V[(size_t)dst].def_pop = gi;
                    V[(size_t)old].uses = 0, V[(size_t)old].def_pop = -1;
                    O[ci].dead = true;
                }
            }
        }
    }

    if (double_hoist) { // the special-prime limbs of every plaintext a group multiplies by (encoded once; a rebuilt plan finds them in place)
        std::vector<int> need_sp;
        for (const Pop &p : O)
            if (!p.dead && p.kind == P_ROTSUM)
                for (int pl : p.plains)
                    if (pl >= 0) need_sp.push_back(pl);
        ensure_special_limbs(need_sp);
    }

    // ---- 3. dataflow depth ------------------------------------------------------------------------------------------
    int max_wave = 0;
    for (Pop &p : O) {
        if (p.dead) continue;
        int w = 0;
        for (int s : p.srcs) {
            const int dp = V[(size_t)s].def_pop;
            if (dp >= 0) w = std::max(w, O[(size_t)dp].wave);
        }
        p.wave = w + 1;
        max_wave = std::max(max_wave, p.wave);
    }

    // ---- 4. steps: per wave, one batch per (kind, level) -----------------------------------------------------------------
    std::vector<std::vector<int>> by_wave((size_t)max_wave + 1);
    for (size_t i = 0; i < O.size(); i++)
        if (!O[i].dead) by_wave[(size_t)O[i].wave].push_back((int)i);
    std::vector<KsItem> h_ks;
    std::vector<MulItem> h_mul;
    std::vector<RsItem> h_rs;
    std::vector<EwItem> h_ew;
    std::vector<SumItem> h_sum;
    std::vector<SumGroup> h_sumg;
    std::vector<SumGroupSrc> h_sumg_srcs;
    double sum_stat[5] = { 0, 0, 0, 0, 0 }; // option trace: sharing of sources between the items of an n-ary sum step
    std::vector<SumSrc> h_srcs;
    std::vector<std::vector<int>> step_pops;
    for (int w = 1; w <= max_wave; w++) {
        std::map<std::tuple<int, int, int>, std::vector<int>> buckets; // (kind, level, target level of opcode 10) -> pops
        for (int pi : by_wave[(size_t)w])
            buckets[std::make_tuple((int)O[(size_t)pi].kind, O[(size_t)pi].level, O[(size_t)pi].target_level)].push_back(pi);
        for (auto &kv : buckets) {
            const PopKind kind = (PopKind)std::get<0>(kv.first);
            // A rotation step's items in key order (option ks_items_fast): SEAL's default key set has 28 Galois elements, so the 64 hops of a
            // convolution's step name each key several times; adjacent in the item table they are adjacent in the fused middle's launch
            // order and read the key out of L2 (fused_ks.hip f_ks_frows_mac_kernel).  The order of a step's items is free: every item names
            // its own source and destination, and a linked consumer step is re-ordered to follow its producer below (4b).  Grouped-digit
            // steps keep program order (their items are grouped by SOURCE, which shares the decomposition).
            if (kind == P_ROT && !c.hybrid() && option(OPT_KS_ITEMS_FAST))
                std::stable_sort(kv.second.begin(), kv.second.end(), [&](int x, int y) { return O[(size_t)x].elt < O[(size_t)y].elt; });
            const bool heavy = kind == P_ROT || kind == P_MULCC || kind == P_RESCALE || kind == P_BOOT || kind == P_ROTSUM;
            const size_t chunk = std::max<size_t>(1, (heavy ? (size_t)max_batch : (size_t)4096) / (size_t)S);
            if (kind == P_ROTSUM) { // whole groups, counted by their rotations
                for (size_t off = 0; off < kv.second.size();) {
                    size_t end = off, items = 0;
                    while (end < kv.second.size() && (items == 0 || items + O[(size_t)kv.second[end]].srcs.size() <= chunk))
                        items += O[(size_t)kv.second[end]].srcs.size(), end++;
                    Step st;
                    st.kind = kind, st.level = std::get<1>(kv.first), st.wave = w, st.count = (int)items;
                    std::vector<int> members(kv.second.begin() + (long)off, kv.second.begin() + (long)end);
                    for (int pi : members) O[(size_t)pi].step = (int)P.steps.size();
                    P.steps.push_back(st);
                    step_pops.push_back(members);
                    off = end;
                }
                continue;
            }
            for (size_t off = 0; off < kv.second.size(); off += chunk) {
                Step st;
                st.kind = kind, st.level = std::get<1>(kv.first), st.target = std::get<2>(kv.first), st.wave = w;
                st.count = (int)std::min(chunk, kv.second.size() - off);
                std::vector<int> members(kv.second.begin() + (long)off, kv.second.begin() + (long)off + st.count);
                for (int pi : members) O[(size_t)pi].step = (int)P.steps.size();
                P.steps.push_back(st);
                step_pops.push_back(members);
            }
        }
    }
    // ---- 4b. chain fusion (plan.hpp Handoff): step B's first phase computed by the last kernel of step A -----------------------------
    // A link needs: B reads A's results and nothing else as its first-phase operand, item for item; no operand expression folded
    // into B; for a multiply, the other operand is A's result too (a square) or older than A.
    P.n_fused = 0;
    std::vector<CtView> h_cont_other;            // filled with stream views in section 6
    std::vector<std::pair<int, int>> cont_other; // (value or -1 for a square, unused) per CONT_MUL item slot, pop order
    std::vector<int> mul_fused_src((size_t)O.size(), -1); // consumer multiply pop -> which operand (0 / 1) comes from the producer
    if (chain_fusion && chain_fusion_supported() && !c.hybrid()) { // (the grouped-digit key switch has no continuation kernels)
        auto def_step_of = [&](int v) -> int { // step defining value v when v is a plain (non-view) pop result, else -1
            const Val &sv = V[(size_t)v];
            if (sv.root != v || sv.def_pop < 0 || O[(size_t)sv.def_pop].dead || O[(size_t)sv.def_pop].dst != v) return -1;
            return O[(size_t)sv.def_pop].step;
        };
        auto wave_of_value = [&](int v) -> int { // wave in which v (or the buffer it views) is complete; 0 for program inputs
            const Val &r = V[(size_t)V[(size_t)v].root];
            return r.def_pop >= 0 ? O[(size_t)r.def_pop].wave : 0;
        };
        for (size_t bi = 0; bi < P.steps.size(); bi++) {
            Step &B = P.steps[bi];
            if (B.kind != P_RESCALE && B.kind != P_BOOT && B.kind != P_MULCC) continue;
            std::vector<int> &bp = step_pops[bi];
            int ai = -1;
            bool ok = true;
            std::vector<int> which(bp.size(), 0);
            for (size_t k = 0; k < bp.size() && ok; k++) {
                const Pop &p = O[(size_t)bp[k]];
                if (p.rs_sum || p.boot_drop >= 0 || (B.kind != P_RESCALE && (p.rs_add >= 0 || p.rs_mul >= 0))) ok = false;
                int found = -1;
                for (size_t w = 0; w < (B.kind == P_MULCC ? 2u : 1u) && ok && found < 0; w++) {
                    const int st = def_step_of(p.srcs[w]);
                    if (st < 0 || (ai >= 0 && st != ai)) continue;
                    const Step &A = P.steps[(size_t)st];
                    const bool kinds = (B.kind == P_RESCALE && A.kind == P_MULCC && A.level == B.level) ||
                                       (B.kind == P_BOOT && ((A.kind == P_RESCALE && A.level - 1 == B.level) || (A.kind == P_ROT && A.level == B.level))) ||
                                       (B.kind == P_MULCC && A.kind == P_RESCALE && A.level - 1 == B.level);
                    if (!kinds || A.fused_consumer >= 0 || A.wave >= B.wave) continue;
                    if (B.kind == P_MULCC) { // the other operand: the same value, or complete before A starts
                        const int other = p.srcs[1 - w];
                        if (other != p.srcs[w] && wave_of_value(other) >= A.wave) continue;
                    }
                    found = st, which[k] = (int)w;
                }
                if (found < 0) ok = false;
                ai = found;
            }
            if (!ok || ai < 0 || step_pops[(size_t)ai].size() != bp.size()) continue;
            // item for item: reorder B's pops to follow A's; every A result must be consumed by exactly one pop of B here
            std::vector<int> order(bp.size(), -1);
            std::vector<int> which2(bp.size(), 0);
            for (size_t k = 0; k < bp.size() && ok; k++) {
                const int src = O[(size_t)bp[k]].srcs[(size_t)which[k]];
                const std::vector<int> &apops = step_pops[(size_t)ai];
                size_t pos = 0;
                while (pos < apops.size() && O[(size_t)apops[pos]].dst != src) pos++;
                if (pos == apops.size() || order[pos] >= 0)
                    ok = false;
                else
                    order[pos] = bp[k], which2[pos] = which[k];
            }
            if (!ok) continue;
            bp = order;
            Step &A = P.steps[(size_t)ai];
            A.fused_consumer = (int)bi;
            A.h.cont = B.kind == P_RESCALE ? CONT_RS : B.kind == P_BOOT ? CONT_BOOT : CONT_MUL;
            if (B.kind == P_MULCC)
                for (size_t k = 0; k < bp.size(); k++) mul_fused_src[(size_t)bp[k]] = which2[k];
            P.n_fused++;
        }
    }
    // steps of one wave are mutually independent: the costliest stays on the main stream, the rest is balanced over
    // (main, auxiliary) when the auxiliary share is worth a fork/join (>= 3 launches)
    if (plan_lanes >= 2) {
        auto cost = [](const Step &st) { return st.kind == P_ROT || st.kind == P_MULCC || st.kind == P_ROTSUM ? 8 : st.kind == P_BOOT ? 5 : st.kind == P_RESCALE ? 3 : 1; };
        for (size_t a = 0; a < P.steps.size();) {
            size_t b = a;
            while (b < P.steps.size() && P.steps[b].wave == P.steps[a].wave) b++;
            std::vector<size_t> idx;
            for (size_t i = a; i < b; i++) idx.push_back(i);
            std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return cost(P.steps[x]) > cost(P.steps[y]); });
            int load[2] = { 0, 0 };
            for (size_t i : idx) {
                const int lane = load[1] < load[0] ? 1 : 0;
                P.steps[i].lane = lane, load[lane] += cost(P.steps[i]);
            }
            if (load[1] < (int)option(OPT_PLAN_AUX_MIN_COST))
                for (size_t i = a; i < b; i++) P.steps[i].lane = 0;
            a = b;
        }
    }
    // ---- 4c. on-line encode: every plaintext register is encoded right before the wave that first reads it, into a window whose
    // blocks are recycled once the last reader's wave is over; the item tables below then see ordinary device pointers -----------
    if (online_encode && !online.items.empty()) {
        const int NW = max_wave + 2;
        std::map<int, std::pair<int, int>> use; // plain register -> (first wave, last wave)
        auto touch = [&](int reg, int w) {
            if (reg < 0) return;
            auto it = use.find(reg);
            if (it == use.end())
                use[reg] = { w, w };
            else
                it->second.first = std::min(it->second.first, w), it->second.second = std::max(it->second.second, w);
        };
        for (size_t si = 0; si < P.steps.size(); si++) {
            const Step &st = P.steps[si];
            // a fused producer evaluates its consumer's folded "+ pt" / "* pt" one step early (Handoff::rs_items)
            const int early = st.fused_consumer >= 0 ? st.wave : -1;
            for (int pi : step_pops[si]) {
                const Pop &p = O[(size_t)pi];
                touch(p.plain, p.wave), touch(p.rs_add, p.wave), touch(p.rs_mul, p.wave);
                for (int pl : p.src_plain) touch(pl, p.wave);
            }
            if (early >= 0)
                for (int pi : step_pops[(size_t)st.fused_consumer]) touch(O[(size_t)pi].rs_add, early), touch(O[(size_t)pi].rs_mul, early);
        }
        std::map<std::pair<int, int>, std::vector<int>> groups; // (first wave, level) -> registers
        for (auto &kv : use) groups[{ kv.second.first, plains.at((size_t)kv.first).level }].push_back(kv.first);
        struct Block { size_t off, size; int release; };
        std::vector<Block> live, freeb;
        size_t high = 0, max_cnt = 0;
        std::vector<EncItem> h_items;
        std::vector<std::pair<size_t, size_t>> placed; // per group: (offset, first item)
        const int chunk = 256;
        int cur_wave = -1;
        for (auto &kv : groups) {
            const int w = kv.first.first, level = kv.first.second;
            if (w != cur_wave) { // blocks whose readers are done are free again
                for (size_t i = 0; i < live.size();)
                    if (live[i].release < w) {
                        freeb.push_back(live[i]);
                        live[i] = live.back(), live.pop_back();
                    } else
                        i++;
                cur_wave = w;
            }
            for (size_t k0 = 0; k0 < kv.second.size(); k0 += (size_t)chunk) {
                const size_t cnt = std::min<size_t>((size_t)chunk, kv.second.size() - k0), size = cnt * (size_t)level * N;
                int release = w;
                for (size_t k = 0; k < cnt; k++) release = std::max(release, use[kv.second[k0 + k]].second);
                size_t off = (size_t)-1;
                for (size_t i = 0; i < freeb.size(); i++)
                    if (freeb[i].size >= size) { // first fit; the remainder stays free
                        off = freeb[i].off;
                        if (freeb[i].size > size)
                            freeb[i].off += size, freeb[i].size -= size;
                        else
                            freeb[i] = freeb.back(), freeb.pop_back();
                        break;
                    }
                if (off == (size_t)-1) off = high, high += size;
                live.push_back({ off, size, release });
                P.enc_groups.push_back({ w, level, (int)h_items.size(), (int)cnt, nullptr });
                placed.push_back({ off, h_items.size() });
                for (size_t k = 0; k < cnt; k++) h_items.push_back(online.items.at(kv.second[k0 + k]));
                max_cnt = std::max(max_cnt, cnt);
            }
            (void)NW;
        }
        DC_HIP_CHECK(vm_malloc(&P.enc_arena, std::max<size_t>(high, 1) * sizeof(u64)));
        P.enc_arena_bytes = high * sizeof(u64);
        P.enc_scratch_bytes = max_cnt * N * sizeof(double2);
        DC_HIP_CHECK(vm_malloc(&P.enc_scratch, std::max<size_t>(P.enc_scratch_bytes, 16)));
        DC_HIP_CHECK(vm_malloc(&P.d_enc_overflow, sizeof(int)));
        DC_HIP_CHECK(hipMemset(P.d_enc_overflow, 0, sizeof(int)));
        P.d_enc_items = upload(h_items);
        size_t gi = 0;
        for (auto &kv : groups)
            for (size_t k0 = 0; k0 < kv.second.size(); k0 += (size_t)chunk, gi++) {
                Plan::EncGroup &g = P.enc_groups[gi];
                g.out = P.enc_arena + placed[gi].first;
                for (int k = 0; k < g.count; k++) plains.at((size_t)kv.second[k0 + (size_t)k]).d = g.out + (size_t)k * (size_t)g.level * N;
            }
        for (auto &kv : online.items)
            if (!use.count(kv.first)) plains.at((size_t)kv.first).d = P.enc_arena; // never read: any valid address
    }
    // ---- 5. lifetimes and pool buffers ---------------------------------------------------------------------------------
    for (const Pop &p : O) {
        if (p.dead) continue;
        V[(size_t)V[(size_t)p.dst].root].def_step = p.step;
        for (int s : p.srcs) {
            Val &r = V[(size_t)V[(size_t)s].root];
            r.last_use = std::max(r.last_use, p.step);
        }
    }
    const size_t buf_elems = (size_t)2 * c.K * N;
    std::vector<u64 *> free_list = P.pool; // every pool buffer is free at plan start
    typedef std::pair<int, u64 *> Rel; // (last_use, buffer)
    std::priority_queue<Rel, std::vector<Rel>, std::greater<Rel>> busy;
    size_t live = 0;
    P.max_live = 0;
    std::vector<std::vector<int>> defs(P.steps.size());
    for (size_t v = 0; v < V.size(); v++)
        if (V[v].root == (int)v && V[v].def_step >= 0 && !V[v].external) defs[(size_t)V[v].def_step].push_back((int)v);
    for (size_t s = 0; s < P.steps.size(); s++) {
        // a buffer is recycled only for values defined in a LATER wave than its last reader (steps of a wave may overlap)
        while (!busy.empty() && P.steps[(size_t)busy.top().first].wave < P.steps[s].wave) {
            free_list.push_back(busy.top().second);
            busy.pop();
            live--;
        }
        for (int v : defs[s]) {
            u64 *b;
            if (!free_list.empty()) {
                b = free_list.back();
                free_list.pop_back();
            } else {
                DC_HIP_CHECK(vm_malloc(&b, buf_elems * (size_t)S * sizeof(u64))); // one block = the value in all S streams
                P.pool.push_back(b);
            }
            V[(size_t)v].buf = b;
            live++;
            P.max_live = std::max(P.max_live, live);
            if (!V[(size_t)v].pinned) busy.push({ std::max(V[(size_t)v].last_use, (int)s), b });
        }
    }
    // ---- 6. device item tables ----------------------------------------------------------------------------------------
    const long ps = (long)c.K * (long)N;
    auto view = [&](int v, int sidx) {
        u64 *b = V[(size_t)V[(size_t)v].root].buf;
        if (!b) {
            fprintf(stderr, "[dacapo_amd] plan: value %d has no buffer (internal error)\n", v);
            abort();
        }
        return CtView{ b + (size_t)sidx * buf_elems, ps };
    };
    size_t need_t = 0, need_d = 0, need_e = 0, need_a = 0, need_m = 0;
    size_t need_bpt = 0, need_bptx = 0;
    int boot_tmax = 0;
    std::vector<std::pair<int, int>> h_boot_pops; // (pop, stream) per boot item; device tables are filled once the arena exists
    std::vector<double> h_boot_ratio;
    P.launches = 0;
    for (size_t s = 0; s < P.steps.size(); s++) {
        Step &st = P.steps[s];
        st.count *= S; // items = pseudo-ops x streams
        const size_t B = (size_t)st.count, l = (size_t)st.level;
        switch (st.kind) {
        case P_ROT:
            st.first = (int)h_ks.size();
            {
                // grouped-digit mode: hops of one source ciphertext share its decomposition (inverse NTT, mod-up, NTT of the raised limbs):
                // bootstrapping's baby steps, a convolution's taps.  slot = index of the source among the step's distinct sources.
                std::map<const u64 *, u32> slot_of;
                for (int pi : step_pops[s])
                    for (int q = 0; q < S; q++) {
                        const CtView sv = view(O[(size_t)pi].srcs[0], q);
                        const u32 slot = c.hybrid() ? slot_of.emplace(sv.p, (u32)slot_of.size()).first->second : 0u;
                        h_ks.push_back(KsItem{ sv, view(O[(size_t)pi].dst, q), O[(size_t)pi].key, O[(size_t)pi].elt, slot });
                    }
                st.unique = (int)slot_of.size();
            }
            break;
        case P_ROTSUM: { // the rotations of every group, adjacent; then one entry per group: dst, first item (relative), item count
            st.first = (int)h_ks.size();
            std::map<const u64 *, u32> slot_of;
            std::vector<KsItem> groups;
            for (int pi : step_pops[s])
                for (int q = 0; q < S; q++) {
                    const Pop &rp = O[(size_t)pi];
                    const CtView dst = view(rp.dst, q);
                    groups.push_back(KsItem{ dst, dst, nullptr, (u32)(h_ks.size() - (size_t)st.first), (u32)rp.srcs.size() });
                    for (size_t k = 0; k < rp.srcs.size(); k++) {
                        const CtView sv = view(rp.srcs[k], q);
                        const u32 slot = slot_of.emplace(sv.p, (u32)slot_of.size()).first->second;
                        const int pl = rp.plains.empty() ? -1 : rp.plains[k];
                        h_ks.push_back(KsItem{ sv, dst, rp.keys[k], rp.elts[k], slot, pl >= 0 ? plains.at((size_t)pl).d : nullptr,
                                               pl >= 0 ? plains.at((size_t)pl).dsp : nullptr });
                    }
                }
            st.unique = (int)slot_of.size();
            st.gfirst = (int)h_ks.size(), st.gcount = (int)groups.size();
            h_ks.insert(h_ks.end(), groups.begin(), groups.end());
            break;
        }
        case P_MULCC:
            st.first = (int)h_mul.size();
            for (int pi : step_pops[s])
                for (int q = 0; q < S; q++)
                    h_mul.push_back(MulItem{ view(O[(size_t)pi].srcs[0], q), view(O[(size_t)pi].srcs[1], q), view(O[(size_t)pi].dst, q) });
            break;
        case P_RESCALE:
            st.first = (int)h_rs.size();
            for (int pi : step_pops[s])
                for (int q = 0; q < S; q++) {
                    const Pop &rp = O[(size_t)pi];
                    RsItem it{ view(rp.srcs[0], q), view(rp.dst, q), 0, 0, nullptr, nullptr };
                    if (rp.rs_sum) {
                        it.first = (int)h_srcs.size(), it.count = (int)rp.srcs.size();
                        for (size_t k = 0; k < rp.srcs.size(); k++) {
                            const int pl = rp.src_plain.empty() ? -1 : rp.src_plain[k];
                            h_srcs.push_back(SumSrc{ view(rp.srcs[k], q), pl >= 0 ? plains.at((size_t)pl).d : nullptr });
                        }
                    }
                    if (rp.rs_add >= 0) it.add = plains.at((size_t)rp.rs_add).d;
                    if (rp.rs_mul >= 0) it.mul = plains.at((size_t)rp.rs_mul).d;
                    h_rs.push_back(it);
                }
            break;
        case P_SUM:
            st.first = (int)h_sum.size();
            if (option(OPT_TRACE)) { // how many of a step's ciphertext reads name a (source list) another item of the step names too
                std::map<std::vector<int>, int> lists;
                std::set<int> distinct;
                size_t terms = 0;
                for (int pi : step_pops[s]) {
                    lists[O[(size_t)pi].srcs]++, terms += O[(size_t)pi].srcs.size();
                    distinct.insert(O[(size_t)pi].srcs.begin(), O[(size_t)pi].srcs.end());
                }
                size_t grouped4 = 0, grouped8 = 0; // ciphertext reads if items with one source list ran 4 / 8 to a thread
                for (auto &kv : lists) grouped4 += kv.first.size() * (size_t)((kv.second + 3) / 4), grouped8 += kv.first.size() * (size_t)((kv.second + 7) / 8);
                sum_stat[0] += (double)st.level * (double)terms, sum_stat[1] += (double)st.level * (double)distinct.size();
                sum_stat[2] += (double)st.level * (double)grouped4, sum_stat[3] += (double)st.level * (double)step_pops[s].size();
                sum_stat[4] += (double)st.level * (double)grouped8;
            }
            for (int pi : step_pops[s])
                for (int q = 0; q < S; q++) {
                    h_sum.push_back(SumItem{ view(O[(size_t)pi].dst, q), (int)h_srcs.size(), (int)O[(size_t)pi].srcs.size() });
                    for (size_t k = 0; k < O[(size_t)pi].srcs.size(); k++) {
                        const int pl = O[(size_t)pi].src_plain.empty() ? -1 : O[(size_t)pi].src_plain[k];
                        h_srcs.push_back(SumSrc{ view(O[(size_t)pi].srcs[k], q), pl >= 0 ? plains.at((size_t)pl).d : nullptr });
                    }
                }
            // The same items as groups that share sources (plan.hpp SumGroup): the output channels of a convolution, the giant steps of a
            // matrix-vector product name the same rotated ciphertexts.  Greedy: an item joins the open group while at least half of its
            // sources are in the group's union already (candidates: the next 64 items of the step, program order keeps relatives close).
            // A source an item names twice gets one union entry per occurrence.  Taken when it saves a fifth of the step's ciphertext reads
            // and leaves option sum_group_min_wgs workgroups.
            if (option(OPT_SUM_GROUP_MIN_WGS) >= 0 && step_pops[s].size() > 1) {
                const std::vector<int> &pops = step_pops[s];
                auto keys_of = [&](int pi) { // (source value, occurrence) keys of an item, in term order
                    std::vector<std::pair<int, int>> ks;
                    std::map<int, int> occ;
                    for (int sv : O[(size_t)pi].srcs) ks.push_back({ sv, occ[sv]++ });
                    return ks;
                };
                std::vector<std::vector<int>> groups;
                std::vector<std::vector<std::pair<int, int>>> unions;
                std::vector<char> used(pops.size(), 0);
                size_t reads_before = 0, reads_after = 0;
                for (size_t a0 = 0; a0 < pops.size(); a0++) {
                    if (used[a0]) continue;
                    used[a0] = 1;
                    std::vector<int> grp{ (int)a0 };
                    std::vector<std::pair<int, int>> uni = keys_of(pops[a0]);
                    std::set<std::pair<int, int>> in_uni(uni.begin(), uni.end());
                    while ((int)grp.size() < kSumGroup) {
                        int best = -1;
                        size_t best_sh = 0;
                        for (size_t b0 = a0 + 1; b0 < pops.size() && b0 < a0 + 64; b0++) {
                            if (used[b0]) continue;
                            const auto kb = keys_of(pops[b0]);
                            size_t sh = 0;
                            for (auto &kk : kb) sh += in_uni.count(kk);
                            if (sh * 2 >= kb.size() && sh > best_sh) best = (int)b0, best_sh = sh;
                        }
                        if (best < 0) break;
                        used[(size_t)best] = 1, grp.push_back(best);
                        for (auto &kk : keys_of(pops[(size_t)best]))
                            if (in_uni.insert(kk).second) uni.push_back(kk);
                    }
                    for (int m : grp) reads_before += O[(size_t)pops[(size_t)m]].srcs.size();
                    reads_after += uni.size();
                    groups.push_back(grp), unions.push_back(uni);
                }
                const long wgs = (long)(N / 256) * st.level * (long)groups.size() * S;
                if (option(OPT_TRACE) >= 2)
                    fprintf(stderr, "[dacapo_amd] plan:   sum step at level %d: %zu items, %zu ciphertext reads, %zu as %zu groups\n", st.level, pops.size(),
                            reads_before, reads_after, groups.size());
                if (reads_after * 5 <= reads_before * 4 && wgs >= (long)option(OPT_SUM_GROUP_MIN_WGS)) {
                    st.gfirst = (int)h_sumg.size();
                    for (int q = 0; q < S; q++)
                        for (size_t gi = 0; gi < groups.size(); gi++) {
                            SumGroup sg{};
                            sg.first = (int)h_sumg_srcs.size(), sg.count = (int)unions[gi].size(), sg.items = (int)groups[gi].size();
                            std::map<std::pair<int, int>, size_t> pos;
                            for (auto &kk : unions[gi]) {
                                SumGroupSrc gs{};
                                gs.v = view(kk.first, q);
                                pos[kk] = h_sumg_srcs.size();
                                h_sumg_srcs.push_back(gs);
                            }
                            for (size_t m = 0; m < groups[gi].size(); m++) {
                                const Pop &p = O[(size_t)pops[(size_t)groups[gi][m]]];
                                sg.dst[m] = view(p.dst, q);
                                const auto km = keys_of(pops[(size_t)groups[gi][m]]);
                                for (size_t k = 0; k < km.size(); k++) {
                                    const int pl = p.src_plain.empty() ? -1 : p.src_plain[k];
                                    h_sumg_srcs[pos[km[k]]].plain[m] = pl >= 0 ? plains.at((size_t)pl).d : sum_add_only();
                                }
                            }
                            h_sumg.push_back(sg);
                        }
                    st.gcount = (int)h_sumg.size() - st.gfirst;
                }
            }
            break;
        case P_MODRAISE:
            need_d = std::max(need_d, B * 2);
            [[fallthrough]];
        case P_NEG:
        case P_COPY:
            st.first = (int)h_ew.size();
            for (int pi : step_pops[s])
                for (int q = 0; q < S; q++)
                    h_ew.push_back(EwItem{ view(O[(size_t)pi].dst, q), view(O[(size_t)pi].srcs[0], q), view(O[(size_t)pi].srcs[0], q) });
            break;
        case P_MULP:
        case P_ADDP:
            st.first = (int)h_ew.size();
            for (int pi : step_pops[s])
                for (int q = 0; q < S; q++)
                    h_ew.push_back(EwItem{ view(O[(size_t)pi].dst, q), view(O[(size_t)pi].srcs[0], q),
                                           CtView{ plains.at((size_t)O[(size_t)pi].plain).d, 0 } });
            break;
        case P_BOOT:
            st.first = (int)h_boot_pops.size();
            for (int pi : step_pops[s])
                for (int q = 0; q < S; q++) h_boot_pops.push_back({ pi, q });
            need_bpt = std::max(need_bpt, B * l), need_bptx = std::max(need_bptx, B * (size_t)st.target);
            boot_tmax = std::max(boot_tmax, st.target);
            break;
        }
        if (st.kind == P_ROT || st.kind == P_MULCC || st.kind == P_ROTSUM) {
            need_t = std::max(need_t, B * l), need_d = std::max(need_d, B * std::max<size_t>(l, 2));
            need_e = std::max(need_e, B * (c.hybrid() ? (size_t)c.hyb_ext((int)l) : l * l));
            need_a = std::max(need_a, B * 2 * (l + (size_t)c.ksp)), need_m = std::max(need_m, B * 2 * l);
            P.launches += 8;
        } else if (st.kind == P_RESCALE) {
            need_d = std::max(need_d, B * 2), need_m = std::max(need_m, B * 2 * l);
            P.launches += 3;
        } else
            P.launches += st.kind == P_BOOT ? 5 : 1;
    }
    // what every step reads and writes, by pool buffer: the dependencies of the explicitly built graph (capture_plan_dag)
    for (size_t s = 0; s < P.steps.size(); s++) {
        Step &st = P.steps[s];
        st.reads.clear(), st.writes.clear();
        for (int pi : step_pops[s]) {
            const Pop &p = O[(size_t)pi];
            for (int v : p.srcs) st.reads.push_back(V[(size_t)V[(size_t)v].root].buf);
            st.writes.push_back(V[(size_t)V[(size_t)p.dst].root].buf);
        }
        if (st.fused_consumer >= 0 && st.h.cont == CONT_MUL) // the producer's last kernel multiplies by the consumer's OTHER operand
            for (int pi : step_pops[(size_t)st.fused_consumer])
                for (int v : O[(size_t)pi].srcs) st.reads.push_back(V[(size_t)V[(size_t)v].root].buf);
        for (auto *vec : { &st.reads, &st.writes }) {
            std::sort(vec->begin(), vec->end());
            vec->erase(std::unique(vec->begin(), vec->end()), vec->end());
        }
    }
    // opcode 10: every item owns a slot of the zero-encryption arena; the zero-encryptions are made in chunks of items with
    // the same target level at the start of each run (plan_zero_encrypt)
    if (!h_boot_pops.empty()) {
        const size_t nb = h_boot_pops.size(), slot = (size_t)2 * boot_tmax * N;
        DC_HIP_CHECK(vm_malloc(&P.zenc, nb * slot * sizeof(u64)));
        P.zenc_bytes = nb * slot * sizeof(u64);
        std::vector<int> order(nb); // boot items sorted by target level (stable): chunks are ranges of the sorted tables
        for (size_t i = 0; i < nb; i++) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            return O[(size_t)h_boot_pops[(size_t)a].first].target_level < O[(size_t)h_boot_pops[(size_t)b].first].target_level;
        });
        const size_t bc = std::min<size_t>(nb, (size_t)std::max(max_batch, 1));
        const size_t cmax = (size_t)boot_tmax + 1;
        DC_HIP_CHECK(vm_malloc(&P.boot_ue, bc * 3 * cmax * N * sizeof(u64)));
        DC_HIP_CHECK(vm_malloc(&P.boot_tmp, bc * 2 * cmax * N * sizeof(u64)));
        for (int ln = 0; ln < plan_lanes; ln++) {
            DC_HIP_CHECK(vm_malloc(&P.boot_pt[ln], std::max<size_t>(need_bpt, 1) * N * sizeof(u64)));
            DC_HIP_CHECK(vm_malloc(&P.boot_ptx[ln], std::max<size_t>(need_bptx, 1) * N * sizeof(u64)));
        }
        std::vector<BootItem> h_boot(nb);
        std::vector<RsItem> h_brs(nb);
        for (size_t k = 0; k < nb;) { // chunks over the sorted order
            const int t = O[(size_t)h_boot_pops[(size_t)order[k]].first].target_level;
            size_t e = k;
            while (e < nb && e - k < bc && O[(size_t)h_boot_pops[(size_t)order[e]].first].target_level == t) e++;
            P.boot_chunks.push_back({ (int)k, (int)(e - k), t });
            for (size_t j = k; j < e; j++) { // sorted position j <-> item order[j]; zenc slot = item index
                const size_t item = (size_t)order[j];
                u64 *z = P.zenc + item * slot;
                h_brs[j] = RsItem{ CtView{ P.boot_tmp + (j - k) * 2 * (size_t)(t + 1) * N, (long)(t + 1) * (long)N }, CtView{ z, (long)t * (long)N } };
            }
            need_d = std::max(need_d, (e - k) * 2), need_m = std::max(need_m, (e - k) * 2 * (size_t)(t + 1));
            P.launches += 7;
            k = e;
        }
        for (size_t item = 0; item < nb; item++) {
            const Pop &p = O[(size_t)h_boot_pops[item].first];
            const int q = h_boot_pops[item].second;
            // scale of what is decrypted: the (possibly folded) rescale's result when there is one, else the source value
            const double new_scale = V[(size_t)p.dst].scale;
            BootItem bi{ view(p.srcs[0], q), view(p.dst, q), P.zenc + item * slot, 1.0 };
            double src_scale;
            if (p.boot_drop >= 0) { // new_scale = 2^floor(log2(scale after the rescale)) was fixed by the SSA walk
                src_scale = p.boot_src_scale * (double)c.primes[(size_t)p.boot_drop];
            } else
                src_scale = V[(size_t)p.srcs[0]].scale;
            bi.ratio = new_scale / src_scale;
            if (p.rs_sum) {
                bi.first = (int)h_srcs.size(), bi.count = (int)p.srcs.size();
                for (size_t k = 0; k < p.srcs.size(); k++) {
                    const int pl = p.src_plain.empty() ? -1 : p.src_plain[k];
                    h_srcs.push_back(SumSrc{ view(p.srcs[k], q), pl >= 0 ? plains.at((size_t)pl).d : nullptr });
                }
            }
            if (p.rs_add >= 0) bi.add = plains.at((size_t)p.rs_add).d;
            if (p.rs_mul >= 0) bi.mul = plains.at((size_t)p.rs_mul).d;
            h_boot[item] = bi;
        }
        P.d_boot = upload(h_boot), P.d_boot_rs = upload(h_brs);
    }
    P.d_ks = upload(h_ks), P.d_mul = upload(h_mul), P.d_rs = upload(h_rs), P.d_ew = upload(h_ew), P.d_sum = upload(h_sum);
    P.h_ew = h_ew;
    P.d_sum_srcs = upload(h_srcs);
    P.d_sumg = upload(h_sumg), P.d_sumg_srcs = upload(h_sumg_srcs);
    // fused links: the consumer's first-phase buffer (owned by the link, so the two steps may sit on different streams / waves)
    // and, for multiplies, the table of the consumers' other operands in item order
    {
        std::vector<size_t> other_first(P.steps.size(), 0);
        for (size_t ai = 0; ai < P.steps.size(); ai++) {
            Step &A = P.steps[ai];
            if (A.fused_consumer < 0) continue;
            Step &B = P.steps[(size_t)A.fused_consumer];
            const size_t items = (size_t)B.count, limbs = A.h.cont == CONT_RS ? 2 : (size_t)B.level;
            u64 *buf = nullptr;
            DC_HIP_CHECK(vm_malloc(&buf, items * limbs * N * sizeof(u64)));
            P.handoff_bufs.push_back(buf);
            A.h.out = buf, B.h.in = buf;
            A.h.sk = keys.sk;
            if (A.h.cont == CONT_RS) A.h.rs_items = P.d_rs + B.first;
            if (A.h.cont == CONT_MUL) {
                other_first[ai] = h_cont_other.size();
                for (int pi : step_pops[(size_t)A.fused_consumer])
                    for (int q = 0; q < S; q++) {
                        const Pop &mp = O[(size_t)pi];
                        const int w = mul_fused_src[(size_t)pi], other = mp.srcs[(size_t)(1 - w)];
                        h_cont_other.push_back(other == mp.srcs[(size_t)w] ? CtView{ nullptr, 0 } : view(other, q));
                    }
            }
        }
        P.d_cont_other = upload(h_cont_other);
        for (size_t ai = 0; ai < P.steps.size(); ai++)
            if (P.steps[ai].fused_consumer >= 0 && P.steps[ai].h.cont == CONT_MUL) P.steps[ai].h.other = P.d_cont_other + other_first[ai];
    }
    auto alloc = [&](size_t limbs) {
        u64 *d = nullptr;
        DC_HIP_CHECK(vm_malloc(&d, std::max<size_t>(limbs, 1) * N * sizeof(u64)));
        return d;
    };
    for (int ln = 0; ln < 2; ln++) {
        BatchWs &w = P.ws[ln];
        for (void *p : { (void *)w.target, (void *)w.digits, (void *)w.ext, (void *)w.acc, (void *)w.tmp })
            if (p) (void)vm_free(p);
        w = BatchWs{};
        if (ln >= plan_lanes) continue;
        w.target = alloc(need_t), w.digits = alloc(need_d), w.ext = alloc(need_e);
        w.acc = alloc(need_a), w.tmp = alloc(need_m);
    }
    {
        size_t aux_waves = 0;
        for (size_t i = 0; i < P.steps.size(); i++)
            if (P.steps[i].lane == 1 && (i == 0 || P.steps[i - 1].wave != P.steps[i].wave || P.steps[i - 1].lane != 1)) aux_waves++;
        while (P.events.size() < 2 * aux_waves + 2) {
            hipEvent_t e;
            DC_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            P.events.push_back(e);
        }
        if (aux_waves && !aux_stream) DC_HIP_CHECK(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
    }
    P.n_keyswitch *= S, P.n_ntt *= S;
    P.ready = true;
    if (plan_graph) capture_plan(); // part of the (untimed) preparation, like the plan itself
    if (option(OPT_TRACE)) {
        static const char *kn[] = { "rot", "mulcc", "rescale", "sum", "neg", "mulp", "addp", "copy", "boot", "modraise", "rotsum" };
        size_t nsteps[kPopKinds] = { 0 }, nitems[kPopKinds] = { 0 };
        std::map<std::pair<int, int>, std::pair<size_t, size_t>> ks; // (kind, level) -> steps, items
        for (const Step &st : P.steps) {
            nsteps[st.kind]++, nitems[st.kind] += (size_t)st.count;
            if (st.kind == P_ROT || st.kind == P_MULCC || st.kind == P_RESCALE || st.kind == P_BOOT || st.kind == P_ROTSUM) {
                auto &e = ks[{ (int)st.kind, st.level }];
                e.first++, e.second += (size_t)st.count;
            }
        }
        if (option(OPT_TRACE) >= 2) { // the first 400 steps, one line per wave
            int lastw = -1;
            for (size_t i = 0; i < P.steps.size() && i < 400; i++) {
                const Step &st = P.steps[i];
                if (st.wave != lastw) fprintf(stderr, "\n  wave %d:", st.wave);
                fprintf(stderr, " [%s l%d x%d%s]", kn[st.kind], st.level, st.count, st.lane ? " aux" : "");
                lastw = st.wave;
            }
            fprintf(stderr, "\n");
        }
        for (int k = 0; k < kPopKinds; k++)
            if (nsteps[k]) fprintf(stderr, "[dacapo_amd] plan:   %-8s %5zu steps %6zu items\n", kn[k], nsteps[k], nitems[k]);
        {
            size_t gsteps = 0, ggroups = 0, gitems = 0;
            for (const Step &st : P.steps)
                if (st.kind == P_SUM && st.gcount > 0) gsteps++, ggroups += (size_t)st.gcount, gitems += (size_t)st.count;
            fprintf(stderr, "[dacapo_amd] plan:   sums sharing sources: %zu steps run %zu items as %zu groups\n", gsteps, gitems, ggroups);
        }
        fprintf(stderr, "[dacapo_amd] plan:   sums, in limbs: %.0f terms (each 2 ciphertext limbs + at most 1 plaintext limb read), %.0f items (2 limbs written); "
                        "distinct sources per step %.0f; ciphertext reads with equal source lists 4 / 8 to a thread %.0f / %.0f\n",
                sum_stat[0], sum_stat[3], sum_stat[1], sum_stat[2], sum_stat[4]);
        { // who consumes what: candidates for handing a result to its consumer's first phase inside one launch
            std::map<std::tuple<int, int, int, int>, int> edges; // (producer kind, consumer kind, uses, same level) -> count
            std::vector<std::vector<int>> users(V.size());
            for (size_t i = 0; i < O.size(); i++)
                if (!O[i].dead)
                    for (int sv : O[i].srcs) users[(size_t)sv].push_back((int)i);
            for (size_t i = 0; i < O.size(); i++) {
                if (O[i].dead) continue;
                const int d = O[i].dst;
                const int nu = (int)users[(size_t)d].size() + (V[(size_t)d].pinned ? 100 : 0);
                if (users[(size_t)d].empty()) edges[std::make_tuple((int)O[i].kind, -1, nu, 0)]++;
                for (int u : users[(size_t)d]) edges[std::make_tuple((int)O[i].kind, (int)O[(size_t)u].kind, nu, O[(size_t)u].wave - O[i].wave)]++;
            }
            for (auto &kv : edges)
                fprintf(stderr, "[dacapo_amd] plan edge: %-8s -> %-8s uses %3d wave distance %3d : %d\n", kn[std::get<0>(kv.first)],
                        std::get<1>(kv.first) < 0 ? "(none)" : kn[std::get<1>(kv.first)], std::get<2>(kv.first), std::get<3>(kv.first), kv.second);
        }
        for (auto &kv : ks)
            fprintf(stderr, "[dacapo_amd] plan:   %-8s level %2d: %5zu steps %6zu items\n", kn[kv.first.first], kv.first.second,
                    kv.second.first, kv.second.second);
    }
    if (option(OPT_TRACE))
        fprintf(stderr, "[dacapo_amd] plan: %zu ops -> %zu pseudo-ops -> %zu steps (%zu on the auxiliary stream, %zu fused into their producer's last kernel) in %d waves, ~%zu launches, %zu live buffers (%.1f GB pool)\n",
                ops.size(), O.size(), P.steps.size(), (size_t)std::count_if(P.steps.begin(), P.steps.end(), [](const Step &st) { return st.lane == 1; }), P.n_fused,
                max_wave, P.launches, P.max_live, (double)P.pool.size() * buf_elems * 8 / 1e9);
}

// the launches of one step, on stream q, with the scratch of the step's lane
void HEVM::issue_step(const Step &st, hipStream_t q)
{
    Context &c = *ctx;
    Plan &P = plan;
    const BatchWs &w = P.ws[st.lane];
    switch (st.kind) {
    case P_ROT: b_rotate_hops(c, w, P.d_ks + st.first, st.count, st.level, q, st.h, st.unique); break;
    case P_ROTSUM: hyb_rotate_sum(c, w, P.d_ks + st.first, st.count, P.d_ks + st.gfirst, st.gcount, st.level, q, st.unique); break;
    case P_MULCC: b_mul_relin(c, w, P.d_mul + st.first, keys.relin, st.count, st.level, q, st.h); break;
    case P_RESCALE: b_rescale(c, w, P.d_rs + st.first, st.count, st.level, q, P.d_sum_srcs, st.h); break;
    case P_SUM:
        if (st.gcount > 0)
            b_sum_group(c, P.d_sumg + st.gfirst, P.d_sumg_srcs, st.gcount, st.level, q);
        else
            b_sum(c, P.d_sum + st.first, P.d_sum_srcs, st.count, st.level, q);
        break;
    case P_NEG: b_ew(c, EwOp::Neg, P.d_ew + st.first, st.count, 2, 2, st.level, q); break;
    case P_COPY: b_ew(c, EwOp::Copy, P.d_ew + st.first, st.count, 2, 2, st.level, q); break;
    case P_MODRAISE: modraise(c, w.digits, P.h_ew.data() + st.first, st.count, st.target, q, P.d_ew + st.first); break;
    case P_MULP: b_ew(c, EwOp::Mul, P.d_ew + st.first, st.count, 2, 1, st.level, q); break;
    case P_ADDP: b_add_plain(c, P.d_ew + st.first, st.count, st.level, q); break;
    case P_BOOT: plan_boot_step(st.first, st.count, st.level, st.target, st.lane, q, st.h); break;
    }
}

void HEVM::issue_plan(hipStream_t s)
{
    Context &c = *ctx;
    Plan &P = plan;
    if (test_zero_enc) { // TEST HOOK (hevm_test_zero_encryption): Enc(0) := (0, 0), so opcode 10 leaves its re-encoded plaintext in c0
        if (P.zenc) DC_HIP_CHECK(hipMemsetAsync(P.zenc, 0, P.zenc_bytes, s));
    } else
        for (const Plan::BootChunk &bc : P.boot_chunks) plan_zero_encrypt(bc.first, bc.count, bc.target, s);
    // option step_profile = 1: synchronise after every step and attribute wall time to (kind, level, batch size) -- a
    // diagnosis mode (every step then pays a full launch round trip, as the steps of a dependent chain do anyway)
    const bool step_profile = option(OPT_STEP_PROFILE) != 0 && !plan_graph;
    std::map<std::tuple<int, int, int>, std::pair<int, double>> prof;
    double wave_max = 0, wave_sum = 0, crit_path = 0, all_steps = 0;
    size_t waves_seen = 0, wide_waves = 0;
    size_t ev = 0, eg = 0;
    for (size_t a = 0; a < P.steps.size();) {
        size_t b = a;
        bool has_aux = false;
        while (b < P.steps.size() && P.steps[b].wave == P.steps[a].wave) has_aux |= P.steps[b].lane == 1, b++;
        for (; eg < P.enc_groups.size() && P.enc_groups[eg].wave <= P.steps[a].wave; eg++) { // on-line encode of this wave's plaintexts
            const Plan::EncGroup &g = P.enc_groups[eg];
            enc_batch(c, enc_tables, online.d_consts, P.d_enc_items + g.first, g.count, g.level, P.enc_scratch, g.out, P.d_enc_overflow, s);
        }
        if (has_aux) { // fork: the auxiliary stream sees everything the main stream has been given so far
            DC_HIP_CHECK(hipEventRecord(P.events[ev], s));
            DC_HIP_CHECK(hipStreamWaitEvent(aux_stream, P.events[ev], 0));
            ev++;
        }
        for (size_t i = a; i < b; i++) {
            const Step &st = P.steps[i];
            hipStream_t q = st.lane ? aux_stream : s;
            issue_step(st, q);
            if (step_profile) {
                static auto t_prev = std::chrono::steady_clock::now();
                if (i == 0) {
                    DC_HIP_CHECK(hipStreamSynchronize(s));
                    t_prev = std::chrono::steady_clock::now();
                }
                DC_HIP_CHECK(hipStreamSynchronize(q));
                const auto t1 = std::chrono::steady_clock::now();
                const int bucket = st.count <= 1 ? 1 : st.count <= 2 ? 2 : st.count <= 4 ? 4 : st.count <= 8 ? 8 : st.count <= 32 ? 32 : 128;
                auto &e = prof[std::make_tuple((int)st.kind, st.level, bucket)];
                const double dt = std::chrono::duration<double>(t1 - t_prev).count();
                e.first++, e.second += dt;
                wave_max = std::max(wave_max, dt), wave_sum += dt;
                t_prev = t1;
            }
        }
        if (has_aux) { // join
            DC_HIP_CHECK(hipEventRecord(P.events[ev], aux_stream));
            DC_HIP_CHECK(hipStreamWaitEvent(s, P.events[ev], 0));
            ev++;
        }
        if (step_profile) { // what a scheduler with unlimited concurrency inside a wave could reach: the wave's longest step
            crit_path += wave_max, all_steps += wave_sum, waves_seen++, wide_waves += (b - a) > 1;
            wave_max = wave_sum = 0;
        }
        a = b;
    }
    bump_epoch(s);
    if (step_profile) {
        static const char *kn[] = { "rot", "mulcc", "rescale", "sum", "neg", "mulp", "addp", "copy", "boot", "modraise", "rotsum" };
        double total = 0;
        for (auto &kv : prof) total += kv.second.second;
        fprintf(stderr, "[dacapo_amd] step profile (synchronised after every step): %.2f ms in %zu steps\n", total * 1e3, P.steps.size());
        // the dataflow graph's width: if every wave ran its steps perfectly overlapped, the run would take the sum of the waves' longest
        // steps -- the bound on what ANY scheduler of this plan's DAG (more streams, an explicitly built graph) can gain over one queue
        fprintf(stderr, "[dacapo_amd]   %zu waves, %zu of them with more than one step; sum of all steps %.2f ms, sum over waves of the longest step %.2f ms "
                        "(= %.1f %% of the serial time)\n", waves_seen, wide_waves, all_steps * 1e3, crit_path * 1e3, 100.0 * crit_path / all_steps);
        for (auto &kv : prof)
            fprintf(stderr, "[dacapo_amd]   %-8s level %2d  batch<=%-3d  %5d steps  %8.3f ms  %7.2f us/step\n", kn[std::get<0>(kv.first)],
                    std::get<1>(kv.first), std::get<2>(kv.first), kv.second.first, kv.second.second * 1e3,
                    kv.second.second * 1e6 / kv.second.first);
    }
}

// Record the plan's launch sequence (main + auxiliary stream) into a HIP graph.  Nothing executes here; kernel arguments are
// the plan's own device tables and pool buffers, which live as long as the plan does.
void HEVM::capture_plan()
{
    Plan &P = plan;
    if (P.graph_exec) return;
    if (!P.boot_chunks.empty() && (!keys.sk || !keys.pk)) return; // opcode 10 on a VM without the keys: run() reports it, eagerly
    // ROCm 7.2's capture walks the recorded nodes recursively: beyond roughly 1.5e5 of them hipStreamEndCapture overflows its stack
    // (seen with the 783 k-instruction real-bootstrap ResNet in on-line encode mode).  Such plans are issued launch by launch.
    if (P.launches + P.enc_groups.size() * (size_t)(ctx->logN + 4) > 120000) return;
    if (plan_dag && capture_plan_dag()) return;
    hipStream_t s = S();
    DC_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    issue_plan(s);
    DC_HIP_CHECK(hipStreamEndCapture(s, &P.graph));
    DC_HIP_CHECK(hipGraphInstantiate(&P.graph_exec, P.graph, nullptr, nullptr, 0));
}

// option plan_graph = 2: the graph BUILT from the plan's own dependencies.  Every step's launch sequence is captured on ONE stream into a
// scratch graph (a chain of kernel nodes: no stream ever waits on another inside a capture, which is what overflows ROCm 7.2's
// hipStreamEndCapture beyond two streams), its nodes are copied into the plan's graph, and the first of them gets explicit edges: to the last
// kernel of every step that wrote a buffer this step reads (or read / wrote one it writes: the pool recycles buffers), to the previous
// user of its lane's scratch, and -- opcode 10 -- to the zero-encryptions at the head of the run.  No per-wave fork / join: a step waits for
// what it needs and nothing else.  false = a node this builder does not copy turned up (the caller falls back to the two-stream capture).
bool HEVM::capture_plan_dag()
{
    Plan &P = plan;
    if (!P.enc_groups.empty()) return false; // (on-line encode interleaves encoder launches with the waves: captured form only)
    hipStream_t cs = S();
    hipGraph_t main = nullptr;
    DC_HIP_CHECK(hipGraphCreate(&main, 0));
    bool ok = true;
    // captures `issue` on cs and appends its kernels to `main` as a chain hanging off `deps`; returns the chain's last node (nullptr: no launch)
    auto add_unit = [&](const std::function<void(hipStream_t)> &issue, std::vector<hipGraphNode_t> deps) -> hipGraphNode_t {
        DC_HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
        issue(cs);
        hipGraph_t g = nullptr;
        DC_HIP_CHECK(hipStreamEndCapture(cs, &g));
        size_t nn = 0, ne = 0;
        DC_HIP_CHECK(hipGraphGetNodes(g, nullptr, &nn));
        std::vector<hipGraphNode_t> nodes(nn);
        if (nn) DC_HIP_CHECK(hipGraphGetNodes(g, nodes.data(), &nn));
        DC_HIP_CHECK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
        std::vector<hipGraphNode_t> from(ne), to(ne);
        if (ne) DC_HIP_CHECK(hipGraphGetEdges(g, from.data(), to.data(), &ne));
        // a single-stream capture is a chain: order it by its edges
        std::map<hipGraphNode_t, hipGraphNode_t> next;
        std::set<hipGraphNode_t> has_pred;
        for (size_t e = 0; e < ne; e++) next[from[e]] = to[e], has_pred.insert(to[e]);
        hipGraphNode_t cur = nullptr;
        for (hipGraphNode_t n : nodes)
            if (!has_pred.count(n)) {
                if (cur) ok = false; // two roots: not a chain
                cur = n;
            }
        if (ne + 1 != nn && nn) ok = false;
        hipGraphNode_t tail = nullptr;
        deps.erase(std::remove(deps.begin(), deps.end(), (hipGraphNode_t) nullptr), deps.end());
        std::sort(deps.begin(), deps.end());
        deps.erase(std::unique(deps.begin(), deps.end()), deps.end());
        for (size_t k = 0; k < nn && ok && cur; k++) {
            hipGraphNodeType ty;
            DC_HIP_CHECK(hipGraphNodeGetType(cur, &ty));
            hipGraphNode_t nw = nullptr;
            const hipGraphNode_t *dp = tail ? &tail : deps.data();
            const size_t nd = tail ? 1 : deps.size();
            if (ty == hipGraphNodeTypeKernel) {
                hipKernelNodeParams kp;
                DC_HIP_CHECK(hipGraphKernelNodeGetParams(cur, &kp));
                DC_HIP_CHECK(hipGraphAddKernelNode(&nw, main, dp, nd, &kp));
            } else if (ty == hipGraphNodeTypeMemset) {
                hipMemsetParams mp;
                DC_HIP_CHECK(hipGraphMemsetNodeGetParams(cur, &mp));
                DC_HIP_CHECK(hipGraphAddMemsetNode(&nw, main, dp, nd, &mp));
            } else
                ok = false;
            tail = nw;
            auto it = next.find(cur);
            cur = it == next.end() ? nullptr : it->second;
        }
        (void)hipGraphDestroy(g);
        return tail;
    };
    // head of the run: the zero-encryptions of every opcode-10 item (the epoch bump -- the next run's randomness -- is the graph's TAIL, behind
    // both lanes, exactly where issue_plan puts it: any step that samples from d_epoch sees the same epoch in all three execution forms)
    hipGraphNode_t zenc_tail = add_unit([&](hipStream_t s) {
        if (test_zero_enc) {
            if (P.zenc) DC_HIP_CHECK(hipMemsetAsync(P.zenc, 0, P.zenc_bytes, s));
        } else
            for (const Plan::BootChunk &bc : P.boot_chunks) plan_zero_encrypt(bc.first, bc.count, bc.target, s);
    }, {});
    // Every step sits on one of the two scratch lanes, and a lane is a chain (its steps share the batch scratch): a dependency on a step is
    // implied by a dependency on any LATER step of the same lane.  So a step needs at most two edges -- its own lane's tail and the latest
    // step it depends on in the other lane -- and none to the other lane when nothing it touches was touched there since.
    struct Ref { int lane = -1; long seq = -1; hipGraphNode_t node = nullptr; }; // the step that last wrote / read a buffer
    std::map<const u64 *, Ref> writer;
    std::map<const u64 *, Ref> reader[2]; // latest reader per lane
    Ref lane_tail[2] = { Ref{ 0, 0, zenc_tail }, Ref{ 1, 0, zenc_tail } }; // (the head uses the lanes' batch scratch too)
    long seq = 0;
    for (size_t i = 0; i < P.steps.size() && ok; i++) {
        const Step &st = P.steps[i];
        const int other = 1 - st.lane;
        Ref cross; // latest step of the other lane this one must follow
        auto need = [&](const Ref &r) {
            if (r.node && r.lane == other && r.seq > cross.seq) cross = r;
        };
        for (const u64 *b : st.reads) {
            auto it = writer.find(b);
            if (it != writer.end()) need(it->second);
        }
        for (const u64 *b : st.writes) {
            auto it = writer.find(b);
            if (it != writer.end()) need(it->second);
            auto rt = reader[other].find(b);
            if (rt != reader[other].end()) need(rt->second);
        }
        std::vector<hipGraphNode_t> deps{ lane_tail[st.lane].node };
        if (cross.node) deps.push_back(cross.node);
        const hipGraphNode_t tail = add_unit([&](hipStream_t s) { issue_step(st, s); }, deps);
        if (!tail) continue;
        const Ref me{ st.lane, ++seq, tail };
        lane_tail[st.lane] = me;
        for (const u64 *b : st.reads) reader[st.lane][b] = me;
        for (const u64 *b : st.writes) writer[b] = me;
    }
    if (ok) (void)add_unit([&](hipStream_t s) { bump_epoch(s); }, { lane_tail[0].node, lane_tail[1].node });
    if (!ok) {
        (void)hipGraphDestroy(main);
        return false;
    }
    P.graph = main;
    DC_HIP_CHECK(hipGraphInstantiate(&P.graph_exec, P.graph, nullptr, nullptr, 0));
    return true;
}

void HEVM::drop_plan_graph()
{
    Plan &P = plan;
    if (P.graph_exec) (void)hipGraphExecDestroy(P.graph_exec);
    if (P.graph) (void)hipGraphDestroy(P.graph);
    P.graph_exec = nullptr, P.graph = nullptr;
}

void HEVM::run_plan()
{
    if (!plan.ready) build_plan();
    Context &c = *ctx;
    Plan &P = plan;
    cur = 0;
    hipStream_t s = S();
    memset(op_counts, 0, sizeof(op_counts));
    for (const WireOp &op : ops)
        if (op.opcode <= 10) op_counts[op.opcode]++;
    n_keyswitch = P.n_keyswitch, n_ntt = P.n_ntt;
    t_bootstrap = 0.0;
    const long ps = (long)c.K * (long)c.N;
    if (plan_graph) { // the plan is a fixed launch sequence: recorded once (normally by preprocess()), replayed as one graph launch
        capture_plan();
        if (P.graph_exec)
            DC_HIP_CHECK(hipGraphLaunch(P.graph_exec, s));
        else
            issue_plan(s);
    } else
        issue_plan(s);
    DC_HIP_CHECK(hipStreamSynchronize(s)); // the caller's timer stops when run() returns
    if (P.d_enc_overflow) { // on-line encode: the range check preprocess() makes for the pre-encoded pool happens here
        int overflow = 0;
        DC_HIP_CHECK(hipMemcpy(&overflow, P.d_enc_overflow, sizeof(int), hipMemcpyDeviceToHost));
        if (overflow) {
            fprintf(stderr, "[dacapo_amd] encode: coefficient does not fit 120 bits (scale too large)\n");
            abort();
        }
    }
    for (size_t r = 0; r < P.final_val.size() && r < ciphers.size(); r++) {
        const int v = P.final_val[r];
        if (v < 0) continue;
        reg_base[r] = P.vals[(size_t)P.vals[(size_t)v].root].buf;
        ciphers[r].data = reg_base[r] + (size_t)sel * (size_t)2 * c.K * c.N;
        ciphers[r].poly_stride = ps;
        ciphers[r].level = P.vals[(size_t)v].level;
        ciphers[r].scale = P.vals[(size_t)v].scale;
    }
}

} // namespace dacapo
