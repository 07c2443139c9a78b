// oracle/_ref/hevm_wire_ref: the ONE piece of the reference's HEVM path that compiles here on its own -- its wire-format
// header /root/reference/include/hecate/Support/HEVMHeader.h (plain structs, <cstdint> only).  This driver (ours) includes that
// header where it lies and walks a .hevm file with the reference's own struct definitions in the order
// SEAL_HEVM::loadHEVM / loadHeader read it (lib/Runtime/SEAL_HEVM.cpp:202-234), printing what the reference runtime would
// see.  tests/test_host_formats.py compares the dump with dacapo_amd.hevm_asm's reader/writer: that pins scope row a1/a6
// (the bytecode container) against the reference's definitions instead of our restatement of them.
// Test infrastructure only; built by oracle/Makefile when /root/reference is present; the binary (not the header) travels.
#include <cstddef>
#include <cstdio>
#include <fstream>
#include <vector>

#include "hecate/Support/HEVMHeader.h"

int main(int argc, char **argv)
{
    printf("{\"sizeof_HEVMHeader\": %zu, \"sizeof_ConfigBody\": %zu, \"sizeof_HEVMOperation\": %zu, ", sizeof(HEVMHeader), sizeof(ConfigBody),
           sizeof(HEVMOperation));
    printf("\"offsetof_arg_length\": %zu, \"offsetof_res_length\": %zu, \"offsetof_init_level\": %zu, \"default_magic\": %u",
           offsetof(HEVMHeader, config_header) + offsetof(HEVMHeader::ConfigHeader, arg_length),
           offsetof(HEVMHeader, config_header) + offsetof(HEVMHeader::ConfigHeader, res_length), offsetof(ConfigBody, init_level),
           HEVMHeader{}.magic_number);
    if (argc < 2) {
        printf("}\n");
        return 0;
    }
    std::ifstream iff(argv[1], std::ios::binary);
    if (!iff) {
        fprintf(stderr, "cannot open %s\n", argv[1]);
        return 1;
    }
    HEVMHeader header;
    ConfigBody config;
    iff.read((char *)&header, sizeof(HEVMHeader));
    iff.read((char *)&config, sizeof(ConfigBody));
    const size_t na = header.config_header.arg_length, nr = header.config_header.res_length;
    std::vector<uint64_t> arg_scale(na), arg_level(na), res_scale(nr), res_level(nr), res_dst(nr);
    iff.read((char *)arg_scale.data(), na * sizeof(uint64_t));
    iff.read((char *)arg_level.data(), na * sizeof(uint64_t));
    iff.read((char *)res_scale.data(), nr * sizeof(uint64_t));
    iff.read((char *)res_level.data(), nr * sizeof(uint64_t));
    iff.read((char *)res_dst.data(), nr * sizeof(uint64_t));
    std::vector<HEVMOperation> ops(config.num_operations);
    iff.read((char *)ops.data(), ops.size() * sizeof(HEVMOperation));
    const bool complete = (bool)iff;
    iff.peek();
    const bool at_end = iff.eof();
    auto arr = [](const char *name, const std::vector<uint64_t> &v) {
        printf(", \"%s\": [", name);
        for (size_t i = 0; i < v.size(); i++) printf("%s%llu", i ? ", " : "", (unsigned long long)v[i]);
        printf("]");
    };
    printf(", \"magic_number\": %u, \"hevm_header_size\": %u, \"config_body_length\": %llu, \"num_operations\": %llu, "
           "\"num_ctxt_buffer\": %llu, \"num_ptxt_buffer\": %llu, \"init_level\": %llu",
           header.magic_number, header.hevm_header_size, (unsigned long long)config.config_body_length,
           (unsigned long long)config.num_operations, (unsigned long long)config.num_ctxt_buffer,
           (unsigned long long)config.num_ptxt_buffer, (unsigned long long)config.init_level);
    arr("arg_scale", arg_scale), arr("arg_level", arg_level), arr("res_scale", res_scale), arr("res_level", res_level), arr("res_dst", res_dst);
    unsigned long long h = 1469598103934665603ull, count[12] = { 0 }; // FNV-1a over (opcode, dst, lhs, rhs) in order
    for (const HEVMOperation &op : ops) {
        for (uint16_t w : { op.opcode, op.dst, op.lhs, op.rhs }) h = (h ^ w) * 1099511628211ull;
        count[op.opcode < 11 ? op.opcode : 11]++;
    }
    printf(", \"ops_fnv1a\": %llu, \"op_counts\": [", h);
    for (int i = 0; i < 12; i++) printf("%s%llu", i ? ", " : "", count[i]);
    printf("], \"complete\": %s, \"at_end\": %s}\n", complete ? "true" : "false", at_end ? "true" : "false");
    return 0;
}
