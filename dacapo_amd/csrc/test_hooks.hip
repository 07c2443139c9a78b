// TEST HOOKS -- linked into libSEAL_HEVM_hooks.so / libSEAL_HEVM_gw_hooks.so only (csrc/Makefile, exports_hooks.map).  The library a maintainer
// copies to $HECATE/build/lib (libSEAL_HEVM.so) exports the reference's 18 symbols (SEAL_HEVM.cpp:404-504) and the safe extensions; what
// lives here would weaken a deployment if it were reachable there:
//   hevm_init_seeded / hevm_init_seeded_primes   every key expanded from a 64-bit seed: reproducible, hence NOT secret
//   hevm_secret_key                              device pointer to the secret key
//   hevm_test_zero_encryption                    every encryption of zero becomes (0, 0)
// tests/ run on the hooks build (the same objects + this file); bench.py, __graft_entry__.smoke() and INTEGRATION.md use the release build.
#define DC_TEST_HOOKS 1 // include/hevm_abi.h declares the hooks (with default visibility) only under this macro
#include "hevm_vm.hpp"

#include "chacha.hpp"
#include "options.hpp"

#include <stdio.h>

namespace dacapo {

RngKeys rng_keys_from_test_seed(uint64_t seed)
{ // splitmix64 expansion: reproducible, NOT secret (64 bits of entropy at most)
    RngKeys k;
    uint32_t *w = reinterpret_cast<uint32_t *>(&k);
    u64 z = seed;
    for (size_t i = 0; i < sizeof(RngKeys) / 4; i += 2) {
        z += 0x9E3779B97F4A7C15ull;
        u64 x = z;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        w[i] = (uint32_t)x, w[i + 1] = (uint32_t)(x >> 32);
    }
    return k;
}

} // namespace dacapo

using dacapo::HEVM;
static HEVM *V(void *vm)
{ // (as in hevm_vm.hip: every entry point names the VM whose allocations it may create or release)
    HEVM *h = static_cast<HEVM *>(vm);
    dacapo::g_vm_allocs = &h->allocs;
    return h;
}

extern "C" {

void *hevm_init_seeded(int logN, int num_primes, uint64_t seed)
{ // logN / num_primes = 0: options logn / primes (the reference's N = 2^15, 14 primes, SEAL_HEVM.cpp:39-40)
    const int dl = (int)dacapo::option(dacapo::OPT_LOGN), dk = (int)dacapo::option(dacapo::OPT_PRIMES);
    auto vm = new HEVM();
    vm->init_context(logN > 0 ? logN : dl, num_primes > 0 ? num_primes : dk, nullptr);
    vm->generate_keys(dacapo::rng_keys_from_test_seed(seed), true, true, true); // reproducible and therefore insecure
    return vm;
}
void *hevm_init_seeded_primes(int logN, const uint64_t *primes, int num_primes, uint64_t seed)
{ // the same on an explicit chain (each prime = 1 mod 2N, 45..60 bits; other than 60: the generic-width build), e.g. a HEaaN-style mixed one
    auto vm = new HEVM();
    vm->init_context(logN, num_primes, primes);
    vm->generate_keys(dacapo::rng_keys_from_test_seed(seed), true, true, true);
    return vm;
}
const uint64_t *hevm_secret_key(void *vm) { return V(vm)->keys.sk; }
void hevm_test_zero_encryption(void *vm, bool on)
{
    if (on) fprintf(stderr, "[dacapo_amd] TEST HOOK: encryptions of zero are (0, 0) from now on -- this VM offers NO security\n");
    V(vm)->test_zero_enc = on;
    V(vm)->drop_plan_graph(); // the recorded launch sequence contains (or lacks) the zero-encryption launches
}

} // extern "C"
