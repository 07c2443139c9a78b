// Host harness for the library's file parsers, built with AddressSanitizer + UBSan (csrc/Makefile target `host_asan`; never part of the
// product library).  The parsers handle files a caller hands over: `.cst` / `.hevm` (wire_parse.cpp; the reference's loaders
// SEAL_HEVM.cpp:182-234 trust them) and SEAL's serialized parm / pub / sec / relin / gal objects and ciphertexts (seal_serial.cpp;
// SEAL_HEVM.cpp:91-180), whose members may be zlib / Zstandard streams.  Usage:
//     host_parsers_asan <kind> <file> [<constants.cst>]      kind = cst | hevm | hevm-header | seal
// Prints "ok ..." or "rejected: <message>" and exits 0 either way; a sanitizer report (or a crash) is the only non-zero exit.
// `seal` walks the file the way HEVM::load_keys does without a context: the outer object, then -- by trial, each attempt from the start of
// the members -- EncryptionParameters, Ciphertext, Plaintext, and a KSwitchKeys walk (parms_id | dim1 | per entry dim2 | nested PublicKeys).
#include <stdio.h>
#include <string.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "seal_serial.hpp"
#include "wire_parse.hpp"

using namespace dacapo;

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        exit(2);
    }
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

// every word a header claims, read the way the product consumes it: bytewise (the data pointer sits behind a 1-byte field: unaligned)
static void touch_words(const uint64_t *d, uint64_t count)
{
    uint64_t x = 0, w;
    for (uint64_t i = 0; i < count; i++) memcpy(&w, (const uint8_t *)d + 8 * i, 8), x ^= w;
    if (x == 0x123456789abcdefull) puts("");
}

static bool header_only_kind(const std::string &k) { return k == "hevm-header"; }

static int do_seal(const std::vector<uint8_t> &file)
{
    std::vector<uint8_t> owned;
    sealio::Reader in(file.data(), file.size(), "object");
    sealio::Reader members = sealio::open_object(in, owned); // (throws on a foreign / truncated / bomb input)
    const size_t msize = members.left();
    const uint8_t *mbase = members.skip(0);
    std::string parsed;
    auto attempt = [&](const char *name, auto fn) {
        try {
            sealio::Reader m(mbase, msize, name);
            fn(m);
            parsed += std::string(parsed.empty() ? "" : ",") + name;
        } catch (const std::runtime_error &) {
        }
    };
    attempt("params", [](sealio::Reader &m) { (void)sealio::get_params(m); });
    attempt("ciphertext", [](sealio::Reader &m) {
        const uint64_t *d = nullptr;
        const sealio::CtHeader h = sealio::get_ciphertext(m, d);
        touch_words(d, h.size * h.limbs * h.N); // touch every word the header claims: ASan checks the claim
    });
    attempt("plaintext", [](sealio::Reader &m) {
        const uint64_t *d = nullptr;
        const sealio::PtHeader h = sealio::get_plaintext(m, d);
        touch_words(d, h.coeff_count);
    });
    attempt("kswitchkeys", [](sealio::Reader &m) { // KSwitchKeys::load_members as hevm_vm.hip get_kswitch_keys walks it
        (void)m.get<sealio::ParmsId>();
        const uint64_t dim1 = m.get<uint64_t>();
        if (dim1 > (1ull << 20)) m.fail("implausible key count");
        for (uint64_t i = 0; i < dim1; i++) {
            const uint64_t dim2 = m.get<uint64_t>();
            if (dim2 > 64) m.fail("implausible digit count");
            for (uint64_t j = 0; j < dim2; j++) {
                std::vector<uint8_t> o2;
                sealio::Reader pk = sealio::open_object(m, o2);
                const uint64_t *d = nullptr;
                const sealio::CtHeader h = sealio::get_ciphertext(pk, d);
                touch_words(d, h.size * h.limbs * h.N);
            }
        }
    });
    printf("ok seal members=%zu parsed=[%s]\n", msize, parsed.c_str());
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s cst|hevm|hevm-header|seal <file> [constants.cst]\n", argv[0]);
        return 2;
    }
    const std::string kind = argv[1];
    const std::vector<uint8_t> file = slurp(argv[2]);
    try {
        std::string err;
        if (kind == "cst") {
            std::vector<std::vector<double>> buf;
            if (!wire::parse_constants(file.data(), file.size(), buf, err)) return printf("rejected: %s\n", err.c_str()), 0;
            size_t total = 0;
            for (auto &v : buf) total += v.size();
            return printf("ok cst constants=%zu values=%zu\n", buf.size(), total), 0;
        }
        if (kind == "hevm" || kind == "hevm-header") {
            std::vector<std::vector<double>> consts;
            if (argc > 3) {
                const std::vector<uint8_t> c = slurp(argv[3]);
                if (!wire::parse_constants(c.data(), c.size(), consts, err)) return printf("rejected: constants: %s\n", err.c_str()), 0;
            }
            wire::Program pr;
            if (!wire::parse_program(file.data(), file.size(), kind == "hevm-header", consts, pr, err)) return printf("rejected: %s\n", err.c_str()), 0;
            // what load_program does next with the result: index the register files by every operand
            std::vector<char> cipher(pr.cipher_registers, 0), plain((size_t)pr.config.num_ptxt_buffer, 0);
            for (const WireOp &op : pr.ops) {
                if (op.opcode > 10 && (op.opcode < kOpEncodeComplex || op.opcode > kOpSetScale)) continue;
                if (op.opcode == 0 || op.opcode == kOpEncodeComplex) {
                    plain.at(op.dst) = 1;
                    continue;
                }
                cipher.at(op.dst) = 1, cipher.at(op.lhs) = 1;
                if (op.opcode == 6 || op.opcode == 8) cipher.at(op.rhs) = 1;
                if (op.opcode == 7 || op.opcode == 9) plain.at(op.rhs) = 1;
                if (op.opcode == kOpSetScale && !(consts.at(op.rhs).at(0) > 0.0)) return puts("BUG: setscale operand passed validation"), 1;
            }
            for (uint64_t r : pr.res_dst)
                if (!header_only_kind(kind)) cipher.at((size_t)r) = 1;
            return printf("ok hevm args=%zu results=%zu ops=%zu cipher_registers=%zu\n", pr.arg_scale.size(), pr.res_dst.size(), pr.ops.size(), pr.cipher_registers), 0;
        }
        if (kind == "seal") return do_seal(file);
        fprintf(stderr, "unknown kind %s\n", kind.c_str());
        return 2;
    } catch (const std::runtime_error &e) {
        printf("rejected: %s\n", e.what());
        return 0;
    } catch (const std::bad_alloc &) { // an allocation the size checks should have prevented
        puts("BUG: allocation failure (a count reached an allocation unchecked)");
        return 1;
    }
}
