// Batched CKKS encoder on the device (encoder.hip).
#pragma once
#include "kernels.hpp"

namespace dacapo {

struct EncItem {
    size_t src_off; // offset (in doubles) of the source vector in the device constant arena
    u32 len;        // its length; 0 = the all-ones constant of an upscale (EmitHEVM.cpp: lhs 0xFFFF)
    u32 cplx;       // 1: the source holds `len` real parts followed by `len` imaginary parts (extension opcode 16, ckks_boot.py)
    double fix;     // scale / N
};
struct EncTables {
    double2 *roots = nullptr; // [N] exp(2 pi i bitrev(k) / 2N), as HostEncoder builds it (long double sin/cos)
    u32 *slot_map = nullptr;  // [N] CKKSEncoder::matrix_reps_index_map_
};

// P plaintexts of `level` primes: scratch [P][N] complex, out [P][level][N] (NTT form on return).  *d_overflow is set if a
// coefficient does not fit 120 bits.  prime_base: the `level` limbs are those of primes prime_base ... (the special-prime limbs of a plaintext
// that multiplies a rotation inside a lazy sum, option hyb_double_hoist: prime_base = the chain's first special prime, level = their number)
void enc_batch(const Context &c, const EncTables &tb, const double *d_consts, const EncItem *d_items, int P, int level, double2 *scratch,
               u64 *out, int *d_overflow, hipStream_t s, int prime_base = 0);

// decode: v [N] complex (real parts = plaintext coefficients / scale) -> out [N/2] slot values (real parts), on the device
void dec_fft(const Context &c, const EncTables &tb, double2 *v, double *out, hipStream_t s);

} // namespace dacapo
