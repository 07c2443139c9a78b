// The reference's two program files, parsed and bounds-checked on the host: `.cst` (constants) and `.hevm` (bytecode), the formats
// SEAL_HEVM::loadConstants / loadHeader / loadHEVM read (/root/reference/lib/Runtime/SEAL_HEVM.cpp:182-234) with the structs of
// include/hecate/Support/HEVMHeader.h:10-35.  The reference trusts both files (unchecked freads into vectors sized by the file's own
// counts, unchecked register indices at run time); here every count is held against the bytes that are there and every operand against
// the register files it names BEFORE anything is allocated or indexed.  Plain C++ with no device code: the same TU is built into the
// library (hevm_vm.hip calls it and aborts with its message) and, with AddressSanitizer + UBSan, into the host harness that
// tests/test_host_fuzz.py drives over a corpus of hostile files (csrc/Makefile target `host_asan`).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace dacapo {

// wire format of include/hecate/Support/HEVMHeader.h:10-35 (little endian, natural alignment)
struct WireHeader {
    uint32_t magic_number;
    uint32_t hevm_header_size;
    uint64_t arg_length;
    uint64_t res_length;
};
struct WireConfigBody {
    uint64_t config_body_length;
    uint64_t num_operations;
    uint64_t num_ctxt_buffer;
    uint64_t num_ptxt_buffer;
    uint64_t init_level;
};
struct WireOp {
    uint16_t opcode, dst, lhs, rhs;
};
// extension opcodes (dacapo_amd/hevm_asm.py OP_ENCODE_COMPLEX ...): not emitted by the reference's compiler, skipped by its VMs
constexpr uint16_t kOpEncodeComplex = 16, kOpConj = 17, kOpModRaise = 18, kOpSetScale = 19;
static_assert(sizeof(WireHeader) == 24 && sizeof(WireConfigBody) == 40 && sizeof(WireOp) == 8, "HEVM wire format");

namespace wire {

// SEAL_HEVM.cpp:182-200: i64 count | count x (i64 length | length x f64).  false + `err` on a truncated or implausible file.
bool parse_constants(const void *data, size_t len, std::vector<std::vector<double>> &buffer, std::string &err);

struct Program {
    WireHeader header{};
    WireConfigBody config{};
    std::vector<uint64_t> arg_scale, arg_level, res_scale, res_level, res_dst;
    std::vector<WireOp> ops;   // empty when header_only
    size_t cipher_registers = 0; // registers the program names: max(arguments + results, num_ctxt_buffer, every operand + 1)
};
// SEAL_HEVM.cpp:202-234 (header_only: loadHeader, :202-217).  `constants` = what parse_constants produced (setscale's operand must exist).
bool parse_program(const void *data, size_t len, bool header_only, const std::vector<std::vector<double>> &constants, Program &out,
                   std::string &err);

} // namespace wire
} // namespace dacapo
