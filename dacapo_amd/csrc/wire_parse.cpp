// .cst / .hevm parsing and validation (wire_parse.hpp).  Host-only C++.
#include "wire_parse.hpp"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

namespace dacapo {
namespace wire {

static std::string fmt(const char *f, long long a = 0, long long b = 0, long long c = 0)
{
    char buf[256];
    snprintf(buf, sizeof buf, f, a, b, c);
    return buf;
}

bool parse_constants(const void *data, size_t len, std::vector<std::vector<double>> &buffer, std::string &err)
{
    const char *p = (const char *)data, *end = p + len;
    if ((size_t)(end - p) < 8) return err = "truncated .cst file", false;
    int64_t count;
    memcpy(&count, p, 8), p += 8;
    if (count < 0 || (uint64_t)count > (uint64_t)(end - p) / 8) // every vector needs at least its 8-byte length
        return err = fmt(".cst file: implausible constant count %lld", (long long)count), false;
    buffer.assign((size_t)count, {});
    for (int64_t i = 0; i < count; i++) {
        if ((size_t)(end - p) < 8) return err = "truncated .cst file", false;
        int64_t veclen;
        memcpy(&veclen, p, 8), p += 8;
        if (veclen < 0 || (uint64_t)veclen > (uint64_t)(end - p) / 8)
            return err = fmt("truncated .cst file (constant %lld claims %lld values)", (long long)i, (long long)veclen), false;
        buffer[(size_t)i].resize((size_t)veclen);
        if (veclen) memcpy(buffer[(size_t)i].data(), p, (size_t)veclen * 8);
        p += veclen * 8;
    }
    return true;
}

bool parse_program(const void *data, size_t len, bool header_only, const std::vector<std::vector<double>> &constants, Program &out,
                   std::string &err)
{
    const char *p = (const char *)data, *end = p + len;
    bool short_file = false;
    auto take = [&](void *dst, size_t n) {
        if (short_file || (size_t)(end - p) < n) {
            short_file = true;
            return;
        }
        if (n) memcpy(dst, p, n);
        p += n;
    };
    take(&out.header, sizeof(out.header));
    take(&out.config, sizeof(out.config));
    if (short_file) return err = "truncated .hevm file", false;
    if (out.header.magic_number != 0x4845564D) return err = fmt("bad .hevm magic 0x%llx", (long long)out.header.magic_number), false;
    const uint64_t na = out.header.arg_length, nr = out.header.res_length, left = (uint64_t)(end - p);
    if (na > left / 16 || nr > left / 24 || out.config.num_operations > left / sizeof(WireOp) || out.config.num_ctxt_buffer > 65536 ||
        out.config.num_ptxt_buffer > 65536) // operands are 16-bit register numbers
        return err = ".hevm header claims more arguments / results / operations / registers than the file can hold", false;
    out.arg_scale.resize((size_t)na), out.arg_level.resize((size_t)na);
    out.res_scale.resize((size_t)nr), out.res_level.resize((size_t)nr), out.res_dst.resize((size_t)nr);
    take(out.arg_scale.data(), (size_t)na * 8), take(out.arg_level.data(), (size_t)na * 8);
    take(out.res_scale.data(), (size_t)nr * 8), take(out.res_level.data(), (size_t)nr * 8), take(out.res_dst.data(), (size_t)nr * 8);
    if (short_file) return err = "truncated .hevm file", false;
    size_t nct = (size_t)(na + nr);
    out.ops.clear();
    if (!header_only) {
        out.ops.resize((size_t)out.config.num_operations);
        take(out.ops.data(), out.ops.size() * sizeof(WireOp));
        if (short_file) return err = "truncated .hevm file", false;
        nct = std::max<size_t>(nct, (size_t)out.config.num_ctxt_buffer);
        const size_t nplain = (size_t)out.config.num_ptxt_buffer;
        // Operand validation, once: the reference indexes its register vectors unchecked (SEAL_HEVM.cpp:268-334); here a program
        // may name cipher registers beyond num_ctxt_buffer (the file grows with them) but never a plaintext register that does not exist.
        for (const WireOp &op : out.ops) {
            if (op.opcode > 10 && (op.opcode < kOpEncodeComplex || op.opcode > kOpSetScale)) continue;
            if (op.opcode == kOpSetScale) { // its operand is a constant that must exist NOW (the run path indexes it unchecked), hold a value, and be a scale
                if (op.rhs >= constants.size() || constants[op.rhs].empty() || !(constants[op.rhs][0] > 0.0) || !std::isfinite(constants[op.rhs][0]))
                    return err = fmt(".hevm: setscale needs constant %lld of %lld to hold a finite positive scale (load the constants before the program)",
                                     (long long)op.rhs, (long long)constants.size()),
                           false;
            }
            if (op.opcode == 0 || op.opcode == kOpEncodeComplex) {
                if (op.dst >= nplain) return err = fmt(".hevm: encode into plaintext register %lld of %lld", (long long)op.dst, (long long)nplain), false;
                continue;
            }
            nct = std::max<size_t>(nct, (size_t)std::max(op.dst, op.lhs) + 1);
            if (op.opcode == 6 || op.opcode == 8) nct = std::max<size_t>(nct, (size_t)op.rhs + 1);
            if ((op.opcode == 7 || op.opcode == 9) && op.rhs >= nplain)
                return err = fmt(".hevm: opcode %lld reads plaintext register %lld of %lld", (long long)op.opcode, (long long)op.rhs, (long long)nplain), false;
        }
        for (uint64_t r : out.res_dst) {
            if (r >= 65536) return err = fmt(".hevm: result register %lld outside the 16-bit register space", (long long)r), false;
            nct = std::max<size_t>(nct, (size_t)r + 1);
        }
    }
    out.cipher_registers = nct;
    return true;
}

} // namespace wire
} // namespace dacapo
