/*
 * hevm_abi.h -- the drop-in boundary: the 18 extern "C" symbols that the reference's libSEAL_HEVM.so exports
 * (/root/reference/lib/Runtime/SEAL_HEVM.cpp:404-504) and that python/hecate/hecate/runner.py:28-71 binds with
 * ctypes.  dacapo_amd/lib/libSEAL_HEVM.so exports exactly these names with these signatures; dropping it into
 * $HECATE/build/lib makes `hc-test <mode> <wl> <bench> SEAL CPU` run on the MI355X.
 *
 * Contract kept from the reference: opaque VM handle allocated with `new` and never freed (no destroy symbol);
 * no error returns -- failures abort the process (SEAL_HEVM.cpp:295,327,496); caller owns every double*;
 * not thread-safe; `run` returns only when the program has finished ON THE DEVICE (the caller stops its timer on
 * return, examples/tests/ResNet.py:109-111).
 */
#ifndef HEVM_ABI_H
#define HEVM_ABI_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: what this header declares is what it exports */
#endif

/* SEAL_HEVM.cpp:405  -- runner.py:35 declares and passes (char*, bool); the 2nd argument is accepted and ignored */
void *initFullVM(char *dir, bool device);
/* SEAL_HEVM.cpp:410 */
void *initClientVM(char *dir);
/* SEAL_HEVM.cpp:415 */
void *initServerVM(char *dir);
/* SEAL_HEVM.cpp:421  -- writes parm/pub/sec/relin/gal ".seal" files in Microsoft SEAL 4.0's binary serialization
 * (SEAL_HEVM.cpp:55-88).  Format-compatible BY CONSTRUCTION, not yet verified against SEAL: the layout (16-byte header, parms_id hash,
 * Galois key index, nested ciphertext objects) follows SEAL 4.0's published serialization and is cross-checked by an independent
 * reader / writer (oracle/seal_format.py), but no file produced by SEAL itself has been read here and none written here has been read
 * by SEAL (SEAL is neither vendored in the reference nor installed; tests/test_seal_diff.py runs where it is).  Keys and all
 * encryption randomness come from ChaCha20 keyed by 512 bits of getrandom(2); the call aborts if that fails.
 * option seal_compr = 0 none (default) | 1 zlib | 2 zstd selects the compr_mode of the written files (DACAPO_HEVM_OPTIONS also takes
 * the names: seal_compr=zlib). */
void create_context(char *dir);
/* SEAL_HEVM.cpp:424 */
void load(void *vm, char *constant, char *vmfile);
/* SEAL_HEVM.cpp:431  -- the reference takes a std::istream*; runner.py:213 passes a path string (broken upstream).
 * Here `is` is treated as a NUL-terminated path to the .hevm file whose header is read. */
void loadClient(void *vm, void *is);
/* SEAL_HEVM.cpp:439 */
void encrypt(void *vm, int64_t i, double *dat, int len);
/* SEAL_HEVM.cpp:446  -- writes N/2 doubles */
void decrypt(void *vm, int64_t i, double *dat);
/* SEAL_HEVM.cpp:458 */
void decrypt_result(void *vm, int64_t i, double *dat);
/* SEAL_HEVM.cpp:464 */
int64_t getResIdx(void *vm, int64_t i);
/* SEAL_HEVM.cpp:470  -- pointer to this runtime's ciphertext register descriptor (struct hevm_ctxt below) */
void *getCtxt(void *vm, int64_t id);
/* SEAL_HEVM.cpp:475 */
void preprocess(void *vm);
/* SEAL_HEVM.cpp:479 */
void run(void *vm);
/* SEAL_HEVM.cpp:483 */
int64_t getArgLen(void *vm);
/* SEAL_HEVM.cpp:487 */
int64_t getResLen(void *vm);
/* SEAL_HEVM.cpp:491 */
void setDebug(void *vm, bool enable);
/* SEAL_HEVM.cpp:495  -- the reference asserts; here the VM is always on the GPU and the call is a no-op */
void setToGPU(void *vm, bool ongpu);
/* SEAL_HEVM.cpp:498 */
void printMem(void *vm);

/* What getCtxt returns (the reference returns seal::Ciphertext*): a ciphertext register resident in HBM. */
struct hevm_ctxt {
    uint64_t *data;     /* device pointer: [2][capacity][N] uint64 limbs, NTT domain */
    int64_t poly_stride; /* elements between the two polynomials */
    int32_t level;      /* number of RNS primes currently in the ciphertext (HEVM "level") */
    int32_t reserved;
    double scale;
};

/* ---- extensions (not in the reference) ------------------------------------------------------------------ */
/* create_context + initFullVM without the files: parameters and the reference's key set (secret, public, relinearisation, default
 * Galois keys) generated in HBM from the operating system's randomness (getrandom), nothing written to disk.  logN / num_primes = 0 take
 * the reference's hard-coded N = 2^15, 14 primes (SEAL_HEVM.cpp:39-40). */
void *hevm_init_fresh(int logN, int num_primes);
/* ... on an explicit prime chain (every prime = 1 mod 2N, 45..60 bits, the last `ks_special` of them special): e.g. a HEaaN-style mixed
 * chain, 60-bit base and special primes around 51-bit rescale primes (HEAAN_HEVM.cpp:55-56, profiled_HEAAN_GPU.json).  Primes narrower
 * than 60 bits need the generic-width build of the library (libSEAL_HEVM_gw.so); the default build aborts with a message. */
void *hevm_init_fresh_primes(int logN, const uint64_t *primes, int num_primes);
#ifdef DC_TEST_HOOKS
/* TEST HOOKS -- exported by libSEAL_HEVM_hooks.so / libSEAL_HEVM_gw_hooks.so only (csrc/test_hooks.hip; the builds tests/ load).  The
 * release libraries do not contain them: seeded keys are not secret, the secret-key pointer and zero encryptions have no place in a
 * deployment.
 * hevm_init_seeded(_primes): hevm_init_fresh(_primes) with all randomness expanded from the 64-bit `seed` (reproducible runs). */
void *hevm_init_seeded(int logN, int num_primes, uint64_t seed);
void *hevm_init_seeded_primes(int logN, const uint64_t *primes, int num_primes, uint64_t seed);
/* device pointer to the secret key [K][N] */
const uint64_t *hevm_secret_key(void *vm);
/* INSECURE: while on, every encryption of zero (encrypt(), opcode 10) is the pair (0, 0), so a ciphertext is
 * (plaintext, 0) and opcode 10's deterministic half -- decrypt, decode, re-encode (SEAL_HEVM.cpp:328-333) -- can be compared
 * limb by limb with the oracle.  Prints a warning when switched on. */
void hevm_test_zero_encryption(void *vm, bool on);
#endif
/* the kernel-level context (dc_context*, include/dacapo_ckks.h) behind a VM */
void *hevm_context(void *vm);
/* device pointers to PUBLIC key material: relin key, galois key for `elt` (NULL if absent), public key */
const uint64_t *hevm_relin_key(void *vm);
const uint64_t *hevm_galois_key(void *vm, uint32_t elt);
const uint64_t *hevm_public_key(void *vm);
/* Key replication over VM replicas (one VM per GPU): the VM's key buffers in a canonical order (secret, public,
 * relinearisation, Galois keys by ascending element) -- device pointers and sizes in 64-bit words; returns their number (call with
 * cap = 0 to size the arrays).  hevm_key_digest: a 64-bit digest of all of them, computed on the device.  hevm_keys_replaced: call
 * after overwriting the buffers from outside (bench.py --broadcast-keys: one flat RCCL broadcast per buffer from GPU 0). */
int hevm_key_buffers(void *vm, uint64_t **ptrs, uint64_t *words, int cap);
uint64_t hevm_key_digest(void *vm);
void hevm_keys_replaced(void *vm);
/* device pointer + level + scale of plaintext register i after preprocess() */
const uint64_t *hevm_plain(void *vm, int64_t i, int32_t *level, double *scale);
/* option "hyb_double_hoist": device pointer to plaintext register i's limbs over the chain's special primes [ks_special][N] (NTT form) once
 * the plan has encoded them -- the registers that multiply a rotation inside a lazy sum -- else NULL.  Test infrastructure, like hevm_plain. */
const uint64_t *hevm_plain_special(void *vm, int64_t i);
/* load a program from memory images of the .cst / .hevm files */
void hevm_load_mem(void *vm, const void *cst, uint64_t cst_len, const void *hevm, uint64_t hevm_len);
/* per-opcode launch statistics of the last run(): counts[11], NTT-equivalents executed */
void hevm_last_run_stats(void *vm, int64_t *op_counts /*[11]*/, int64_t *keyswitches, int64_t *ntts);
/* option "hyb_lazy_sum" (grouped-digit mode, off by default): the rotate instructions the last run()'s plan executed as lazy sums -- the
 * accumulators of a group's key switches added in the raised basis, ONE division by P per group.  out = [n_0, op ...,
 * n_1, op ...]: per group its size and its rotations' instruction indices.  Returns the length of that list (written if cap suffices), 0 without
 * groups, -1 before the first run() and under option "plan" = 0 (the loop executes every rotate on its own).  Test infrastructure: oracle/oracle.py OracleVM.set_lazy_groups replays exactly these groups. */
int64_t hevm_plan_lazy_groups(void *vm, int32_t *out, int64_t cap);
/* Throughput mode: run `n` independent ciphertext streams of the same program side by side (shared keys and
 * plaintexts; every step of the batched plan processes all streams in one launch sequence).  Call before load();
 * encrypt / decrypt / decrypt_result / getCtxt then address the stream chosen with hevm_select_stream. */
void hevm_set_streams(void *vm, int n);
void hevm_select_stream(void *vm, int s);
/* wall seconds the last run() spent inside opcode 10 (decrypt / re-encode / encrypt) */
double hevm_last_run_bootstrap_seconds(void *vm);
/* HBM bytes held for the loaded program's plaintexts (pre-encoded pool, or constants + encode window with
 * option online_encode = 1: plaintexts encoded at use, HEAAN_HEVM.cpp:266-281) */
uint64_t hevm_plaintext_bytes(void *vm);
/* Frees a VM: its keys, registers, plaintext pool, plan, streams and graph.  The reference's ABI has no such symbol (its VM handles are
 * allocated with `new` and never freed, SEAL_HEVM.cpp:404-419): a host that creates VMs repeatedly can call this; the handle (and the
 * pointers hevm_context / getCtxt returned for it) must not be used afterwards. */
void hevm_destroy(void *vm);
/* Direct Galois keys for the given slot offsets (left rotation = positive), what KeyGenerator::create_galois_keys(steps, ...)
 * makes in SEAL and what the reference's HEaaN runtime loads for its fixed offset list (HEAAN_HEVM.cpp:58-64,124-126).  A
 * rotation by such an offset is then ONE key switch instead of one per non-zero NAF digit (Evaluator::rotate_internal uses a
 * direct key when it exists).  Needs the secret key; call before load()/preprocess() or re-run preprocess() afterwards.
 * The reference's SEAL runtime only ever has the default set (SEAL_HEVM.cpp:82-83): programs run identically without this. */
void hevm_add_rotation_keys(void *vm, const int64_t *offsets, int count);
/* seal::Ciphertext::save / ::load of cipher register `reg` (the reference hands out seal::Ciphertext* through getCtxt "to
 * implement communication", SEAL_HEVM.cpp:463-473): SEAL 4.0 bytes, parms_id of the register's level, its scale. */
void hevm_save_ctxt(void *vm, int64_t reg, const char *path);
void hevm_load_ctxt(void *vm, int64_t reg, const char *path);

/* Run-time options (dacapo_amd/csrc/options.hpp holds the ONE table: names, defaults, meaning).  VM options -- "logn", "primes",
 * "prime_bits", "ks_special", "ks_alpha", "secret_hw", "rot_compose", "plan", "plan_graph", "plan_lanes", "max_batch", "chain_fusion", "host_encoder",
 * "online_encode", "fold_rescale_boot", "hyb_mfma", "hyb_fuse", "seal_compr", "trace", "step_profile" -- are read when a VM (or kernel-level
 * context) is created; launch-shape thresholds -- "small_tile_wgs", "tiny_tile_wgs", "ntt_full_min_limbs", ... -- at every launch.
 * Process-wide, not thread-safe (like the rest of this ABI).  An unknown name aborts with the list of names.  A caller that only knows
 * the reference's 18 symbols sets the same names through the single environment variable DACAPO_HEVM_OPTIONS="name=value,...". */
int hevm_set_option(const char *name, long long value);
long long hevm_get_option(const char *name);
void hevm_reset_options(void);

/* ---- host-only helpers: SEAL 4.0 serialization and the PRNG's block function, no GPU touched --------------- */
/* EncryptionParameters::parms_id of CKKS parameters (poly_modulus_degree, primes[0..count)): BLAKE2b-256 */
void hevm_seal_parms_id(uint64_t poly_modulus_degree, const uint64_t *primes, int count, uint64_t out[4]);
/* EncryptionParameters::save / ::load (compr_mode 0 none, 1 zlib, 2 zstd); load returns the number of primes */
void hevm_seal_save_parms(const char *path, int compr_mode, uint64_t poly_modulus_degree, const uint64_t *primes, int count);
int hevm_seal_load_parms(const char *path, uint64_t *poly_modulus_degree, uint64_t *primes, int capacity);
/* Ciphertext::save / ::load on host limbs [size][limbs][N]; the parms_id written is that of primes[0..limbs).
 * load returns the number of 64-bit words of the data array and copies them when `capacity` allows. */
void hevm_seal_save_ciphertext(const char *path, int compr_mode, uint64_t poly_modulus_degree, const uint64_t *primes, int limbs, int size,
                               int is_ntt, double scale, const uint64_t *data);
int64_t hevm_seal_load_ciphertext(const char *path, uint64_t *poly_modulus_degree, int *limbs, int *size, int *is_ntt, double *scale,
                                  uint64_t parms_id[4], uint64_t *data, uint64_t capacity);
int hevm_seal_zstd_available(void);
/* ChaCha20 block function (RFC 8439 2.3; 64-bit counter in state words 12-13, 64-bit nonce in 14-15) on the host and,
 * for the parity test of the samplers' code path, on the GPU */
void hevm_chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t nonce, uint32_t out[16]);
void hevm_chacha20_blocks_device(const uint32_t key[8], uint64_t counter, uint64_t nonce, int blocks, uint32_t *out_host);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
