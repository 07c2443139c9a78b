// ONE table of run-time options for the whole library (round 4; rounds 1-3 had 31 getenv() knobs, most latched in function-local statics).
//
// The reference's runtime has no options at all: N = 2^15 / 14 primes are constants of SEAL_HEVM.cpp:39-40, and its caller can only reach
// the 18 symbols of SEAL_HEVM.cpp:404-504.  Everything this library adds is therefore optional and lives here, in two groups:
//   * VM options   -- read once, when a VM is created (initFullVM / initClientVM / initServerVM / create_context / hevm_init_seeded): ring,
//                     chain, key-switching mode, execution mode.  A VM keeps what it was created with.
//   * launch shapes -- the thresholds at which a batched launch changes its kernel form (tile geometry, single-crossing NTT, merged
//                     key-switch phases, paired n-ary sum), in workgroups / limbs of the launch, i.e. keyed by (N, level, batch) through
//                     the launch's size.  Read at every launch: a test flips them in-process (no forked children, nothing latched).
// Set through the C entry point hevm_set_option(name, value) (include/hevm_abi.h) -- process-wide, not thread-safe, like the rest of the
// ABI -- or, for a caller that only knows the reference's 18 symbols, through the single environment variable
//     DACAPO_HEVM_OPTIONS="name=value,name=value,..."
// parsed once, the first time any option is read.  Unknown names abort with the list of known ones.
#pragma once

namespace dacapo {

enum Opt : int {
    // ---- VM options ----------------------------------------------------------------------------------------------------------------
    OPT_LOGN,              // ring degree of create_context / init*VM (reference: 15)
    OPT_PRIMES,            // primes in the key-level chain (reference: 14)
    OPT_PRIME_BITS,        // width of the chain's primes, 45..60 (other than 60: libSEAL_HEVM_gw.so)
    OPT_KS_SPECIAL,        // grouped-digit key switching: special primes (1 = SEAL's scheme)
    OPT_KS_ALPHA,          // ... data primes per digit (0 = same as ks_special)
    OPT_SECRET_HW,         // Hamming weight of the ternary secret (0 = SEAL's uniform ternary)
    OPT_ROT_COMPOSE,       // rotations without a direct key: shortest sum of offsets that have one (0: SEAL's NAF over the power-of-two keys)
    OPT_PLAN,              // 1: batched execution plan; 0: the reference's loop, one instruction at a time
    OPT_PLAN_GRAPH,        // replay the plan as one HIP graph: 1 captured from two streams (fork / join per wave), 2 built from the plan's own dependencies, 0 off
    OPT_PLAN_LANES,        // streams the plan's independent steps are spread over (1 or 2)
    OPT_PLAN_AUX_MIN_COST, // a wave's auxiliary-stream share must be worth a fork / join: at least this many cost units (key switch 8, opcode 10 5, rescale 3, element-wise 1)
    OPT_MAX_BATCH,         // items per heavy batched step
    OPT_CHAIN_FUSION,      // producer's last kernel runs the consumer's first phase
    OPT_HOST_ENCODER,      // encode / decode on the host (comparison only)
    OPT_ONLINE_ENCODE,     // encode plaintexts at use instead of pre-encoding the pool
    OPT_FOLD_RESCALE_BOOT, // opcode 10 absorbs a rescale that only it consumes (changes what is computed homomorphically: off)
    OPT_HYB_MFMA,          // grouped-digit base conversions on the matrix cores (0: vector kernels)
    OPT_HYB_FUSE,          // grouped-digit key switch: 2 (default) fused sequence, base conversions as pre-scaled matrix-core launches; 1 fused, conversions in the transforms' loaders; 0 round 3's sequence
    OPT_HYB_LAZY_SUM,      // grouped-digit plan: rotations (the last hop of a rotate instruction) whose single-use results are bare terms of one sum share ONE mod-down (changes the rounding: off; hevm_plan_lazy_groups names the groups; 2: same sums through per-item accumulators and a separate adding kernel)
    OPT_HYB_DOUBLE_HOIST,  // with hyb_lazy_sum: a rotation multiplied by a plaintext before it joins the sum is a member too -- the product is taken in the raised basis (the plaintext's special-prime limbs are encoded at preprocess) and the group still has ONE mod-down (changes the rounding like hyb_lazy_sum: off)
    OPT_SEAL_COMPR,        // compression of written .seal files: 0 none, 1 zlib, 2 zstd
    OPT_TRACE,             // plan statistics on stderr (2: per wave)
    OPT_STEP_PROFILE,      // per-step timing of an un-graphed plan on stderr
    // ---- launch shapes ---------------------------------------------------------------------------------------------------------------
    OPT_SMALL_TILE_WGS,          // launches below this many 2048-coefficient tiles take the 1024-coefficient (radix-4) geometry (-1: the per-ring table of ntt_tile.hpp)
    OPT_TINY_TILE_WGS,           // launches of at most this many 512-coefficient tiles take the one-butterfly geometry
    OPT_WIDE_TILE_WGS,           // the key switch's lift launches (base change + forward COLS phase) of at least this many 4096-coefficient tiles take the radix-16 geometry (-1: never)
    OPT_ROWS_PRIME_MAJOR,        // rings of at least 2^this: a ROWS-phase launch walks its limbs prime by prime (limbs of one prime adjacent in launch order: the prime's twiddle tiles, as large as the data at N >= 2^16, are read out of L2 by all but the first); 0 = never
    OPT_NTT_FULL_MIN_LIMBS,      // N = 2^15: forward launches of at least this many limbs take the single-crossing kernel (0 = never)
    OPT_NTT_FULL_INV_MIN_LIMBS,  // ... inverse launches
    OPT_NTT_FULL_PERSIST,        // workgroups of its persistent grid (-1: one per CU; 0: one workgroup per limb)
    OPT_NTT_FULL_INV_PERSIST,    // ... inverse (-1: same as forward)
    OPT_NTT_FULL_PAIRS,          // twiddle pairs in its forward passes A and B
    OPT_NTT_FULL_INV_PAIRS,      // twiddle pairs in its inverse passes B and A
    OPT_COLS_PAIRS,              // twiddle pairs in the forward COLS tiles of the fused key-switch / rescale phases (60-bit build)
    OPT_KS_MERGE_SPECIAL_MIN_WGS, // fused key-switch middle: one workgroup row for both special-prime accumulators from this many workgroups
    OPT_KS_MERGE_LIFT_MIN_WGS,   // L2 / L6: share the inverse phase among a source limb's targets while this many workgroups remain
    OPT_KS_ITEMS_FAST,           // fused key-switch middle: items faster than rows in launch order, and a rotation step's items sorted by key (items sharing a key read it out of L2)
    OPT_KS_FUSE_MAC,             // second NTT phase + inner products + first inverse phase in one launch
    OPT_KS_BIG_TILES,            // work (1024-coefficient tiles of lifted digits) from which a key switch takes the large-batch sequence
    OPT_KS_FUSE_MAC_TILES,       // ... and up to which its middle stays one launch
    OPT_SUM_PAIR_MIN_WGS,        // n-ary sum: both polynomials per workgroup while this many workgroups remain
    OPT_SUM_GROUP_MIN_WGS,       // n-ary sums of a step that share sources run up to 8 to a thread (each shared ciphertext limb read once) while this many workgroups remain (-1: never)
    OPT_COUNT
};

long long option(Opt o);

} // namespace dacapo
