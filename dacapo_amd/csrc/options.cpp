// The option table of options.hpp: names, defaults, the one environment variable, and the C entry points hevm_set_option /
// hevm_get_option / hevm_reset_options (include/hevm_abi.h).  Host-only C++.
#include "options.hpp"

#include "../../include/hevm_abi.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace dacapo {

struct OptDef {
    const char *name;
    long long def;
};
// same order as enum Opt.  Launch-shape defaults are measured (profiles/r02_experiments.txt, r03_ntt_full_check.txt, r03_experiments.txt).
static const OptDef kDefs[OPT_COUNT] = {
    { "logn", 15 },
    { "primes", 14 },
    { "prime_bits", 60 },
    { "ks_special", 1 },
    { "ks_alpha", 0 },
    { "secret_hw", 0 },
    { "rot_compose", 0 },
    { "plan", 1 },
    { "plan_graph", 1 },
    { "plan_lanes", 2 },
    { "plan_aux_min_cost", 3 },
    { "max_batch", 64 },
    { "chain_fusion", 1 },
    { "host_encoder", 0 },
    { "online_encode", 0 },
    { "fold_rescale_boot", 0 },
    { "hyb_mfma", 1 },
    { "hyb_fuse", 2 },
    { "hyb_lazy_sum", 0 },
    { "hyb_double_hoist", 0 },
    { "seal_compr", 0 },
    { "trace", 0 },
    { "step_profile", 0 },
    { "small_tile_wgs", -1 },
    { "tiny_tile_wgs", 2000 },
    { "wide_tile_wgs", 2048 },
    { "rows_prime_major", 16 },
    { "ntt_full_min_limbs", 768 },
    { "ntt_full_inv_min_limbs", 768 },
    { "ntt_full_persist", -1 },
    { "ntt_full_inv_persist", -1 },
    { "ntt_full_pairs", 1 },
    { "ntt_full_inv_pairs", 1 },
    { "cols_pairs", 1 },
    { "ks_merge_special_min_wgs", 2048 },
    { "ks_merge_lift_min_wgs", 1024 },
    { "ks_items_fast", 1 },
    { "ks_fuse_mac", 1 },
    { "ks_big_tiles", 4096 },
    { "ks_fuse_mac_tiles", 1LL << 40 },
    { "sum_pair_min_wgs", 1024 },
    { "sum_group_min_wgs", 2048 },
};

static long long g_val[OPT_COUNT];
static bool g_init = false;

static int find(const char *name, size_t len)
{
    for (int i = 0; i < OPT_COUNT; i++)
        if (strlen(kDefs[i].name) == len && !strncmp(kDefs[i].name, name, len)) return i;
    return -1;
}

static void unknown(const char *what, const char *name, size_t len)
{
    fprintf(stderr, "[dacapo_amd] %s: unknown option \"%.*s\"; known:", what, (int)len, name);
    for (int i = 0; i < OPT_COUNT; i++) fprintf(stderr, " %s", kDefs[i].name);
    fprintf(stderr, "\n");
    abort();
}

// value of "name=value": a complete integer (decimal, 0x.., 0..), or one of seal_compr's names; anything else aborts -- "plan=", "max_batch=12abc"
// and "seal_compr=zlib" used to run as 0, 12 and 0 (a mistyped option must not silently run something else)
static long long parse_value(int opt, const char *v, size_t len)
{
    if (opt == OPT_SEAL_COMPR) {
        static const char *const names[] = { "none", "zlib", "zstd" };
        for (int k = 0; k < 3; k++)
            if (strlen(names[k]) == len && !strncmp(names[k], v, len)) return k;
    }
    char buf[32];
    char *end = nullptr;
    if (len == 0 || len >= sizeof buf) goto bad;
    memcpy(buf, v, len), buf[len] = 0;
    {
        const long long r = strtoll(buf, &end, 0);
        if (end != buf && *end == 0) return r;
    }
bad:
    fprintf(stderr, "[dacapo_amd] DACAPO_HEVM_OPTIONS: option \"%s\" needs an integer value%s, got \"%.*s\"\n", kDefs[opt].name,
            opt == OPT_SEAL_COMPR ? " (or none | zlib | zstd)" : "", (int)len, v);
    abort();
}

// Rounds 1-3 read 31 environment variables (DACAPO_HEVM_*, DACAPO_KS_*, DACAPO_NTT_*, ...); they are no longer read.  A script that still
// sets one would silently run the defaults (N = 2^15, SEAL-mode keys): say so once, naming the replacement.
extern "C" char **environ;
static void warn_legacy_environment()
{
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "DACAPO_", 7) != 0) continue;
        if (!strncmp(*e, "DACAPO_HEVM_OPTIONS=", 20) || !strncmp(*e, "DACAPO_AMD_LIB=", 15) || !strncmp(*e, "DACAPO_FORCE_DIST=", 18) ||
            !strncmp(*e, "DACAPO_AMD_HOOKS=", 17)) // (read by the Python package: which build of the library it loads)
            continue;
        const char *eq = strchr(*e, '=');
        fprintf(stderr, "[dacapo_amd] warning: environment variable %.*s is no longer read (round 4 replaced the per-knob variables); use "
                        "DACAPO_HEVM_OPTIONS=\"name=value,...\" or hevm_set_option() -- names: include/hevm_abi.h, csrc/options.hpp\n",
                eq ? (int)(eq - *e) : (int)strlen(*e), *e);
    }
}

static void init()
{
    if (g_init) return;
    g_init = true;
    for (int i = 0; i < OPT_COUNT; i++) g_val[i] = kDefs[i].def;
    warn_legacy_environment();
    const char *e = getenv("DACAPO_HEVM_OPTIONS"); // the only environment variable the library reads
    while (e && *e) {
        const char *end = strchr(e, ',');
        const size_t len = end ? (size_t)(end - e) : strlen(e);
        const char *eq = (const char *)memchr(e, '=', len);
        if (len) {
            if (!eq) unknown("DACAPO_HEVM_OPTIONS (expected name=value)", e, len);
            const int i = find(e, (size_t)(eq - e));
            if (i < 0) unknown("DACAPO_HEVM_OPTIONS", e, (size_t)(eq - e));
            g_val[i] = parse_value(i, eq + 1, len - (size_t)(eq + 1 - e));
        }
        e = end ? end + 1 : nullptr;
    }
}

long long option(Opt o)
{
    init();
    return g_val[o];
}

} // namespace dacapo

extern "C" {

// 0 on success; an unknown name aborts with the list of names (a mistyped option must not silently run the default)
int hevm_set_option(const char *name, long long value)
{
    dacapo::init();
    const int i = dacapo::find(name, strlen(name));
    if (i < 0) dacapo::unknown("hevm_set_option", name, strlen(name));
    dacapo::g_val[i] = value;
    return 0;
}

long long hevm_get_option(const char *name)
{
    dacapo::init();
    const int i = dacapo::find(name, strlen(name));
    if (i < 0) dacapo::unknown("hevm_get_option", name, strlen(name));
    return dacapo::g_val[i];
}

// every option back to its default (DACAPO_HEVM_OPTIONS is not re-read)
void hevm_reset_options(void)
{
    dacapo::init();
    for (int i = 0; i < dacapo::OPT_COUNT; i++) dacapo::g_val[i] = dacapo::kDefs[i].def;
}
}
