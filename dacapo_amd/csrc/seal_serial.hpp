// Microsoft SEAL 4.0 binary serialization, host side: what seal::EncryptionParameters / PublicKey / SecretKey / RelinKeys /
// GaloisKeys / Ciphertext / Plaintext ::save write and ::load accept.  The reference's runtime stores its context as five
// such files (/root/reference/lib/Runtime/SEAL_HEVM.cpp:55-88 create_context, :91-180 loadSEAL / loadClient / loadServer);
// reading and writing the same bytes is what lets a key directory made by the reference (or any SEAL 4.0 program) drive this
// runtime and the other way round.  Format [SEAL-upstream serialization.h/.cpp, encryptionparams.cpp, ciphertext.cpp,
// plaintext.cpp, kswitchkeys.cpp, dynarray.h]:
//
//   every object  = SEALHeader (16 bytes) + members, the members possibly compressed as one zlib / Zstandard stream
//   SEALHeader    = u16 magic 0xA15E | u8 header_size 16 | u8 version_major 4 | u8 version_minor 0 | u8 compr_mode
//                   (0 none, 1 zlib, 2 zstd) | u16 reserved 0 | u64 size (header + stored members, bytes)
//   DynArray<u64> = u64 count | count * u64                                          (nested, own header, compr none)
//   Modulus       = u64 value                                                        (nested, own header)
//   EncryptionParameters = u8 scheme (2 = ckks) | u64 poly_modulus_degree | u64 coeff_modulus_size | Modulus * size |
//                          Modulus plain_modulus (0 for ckks)
//   Ciphertext    = parms_id (4 * u64) | u8 is_ntt_form | u64 size | u64 poly_modulus_degree | u64 coeff_modulus_size |
//                   u64 correction_factor | f64 scale | DynArray data [size][coeff_modulus_size][N]
//                   (a DynArray of half that length followed by a UniformRandomGeneratorInfo is the seed-compressed form of
//                   Serializable<>: never written by the reference, rejected here with a message)
//   PublicKey     = a Ciphertext (size 2, key-level parms_id, scale 1)
//   Plaintext     = parms_id | u64 coeff_count | f64 scale | DynArray data ;  SecretKey = a Plaintext in NTT form, key level
//   KSwitchKeys   = parms_id | u64 dim1 | for each: u64 dim2 | PublicKey * dim2
//                   RelinKeys: dim1 = 1 (s^2), dim2 = decomposition digits;  GaloisKeys: dim1 = N, entry (elt - 1) / 2
//   parms_id      = BLAKE2b-256 of the u64 words { scheme, N, q_0 .. q_{k-1}, plain_modulus }; a ciphertext at l primes
//                   carries the id of the parameters truncated to q_0 .. q_{l-1}, keys the id of the full chain
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <array>
#include <string>
#include <vector>

namespace dacapo {
namespace sealio {

typedef std::array<uint64_t, 4> ParmsId;
enum Compr : uint8_t { COMPR_NONE = 0, COMPR_ZLIB = 1, COMPR_ZSTD = 2 };
constexpr uint16_t kMagic = 0xA15E;
constexpr uint8_t kSchemeCkks = 2;

// RFC 7693, unkeyed
void blake2b(void *out, size_t outlen, const void *in, size_t inlen);
ParmsId parms_id(uint64_t N, const uint64_t *primes, size_t count, uint8_t scheme = kSchemeCkks, uint64_t plain_modulus = 0);

// All failures (truncated / foreign / unsupported input, I/O errors) print a message and abort, like the reference's
// uncaught SEAL exceptions (SEAL_HEVM.cpp is built with -fno-exceptions); `what` names the object for the message.
class Reader {
  public:
    Reader(const uint8_t *p, size_t n, std::string what) : p_(p), end_(p + n), what_(std::move(what)) {}
    template <class T>
    T get()
    {
        T v;
        take(&v, sizeof(T));
        return v;
    }
    void take(void *dst, size_t n);
    const uint8_t *skip(size_t n); // returns the start of the skipped range
    size_t left() const { return (size_t)(end_ - p_); }
    const std::string &what() const { return what_; }
    [[noreturn]] void fail(const char *msg) const;

  private:
    const uint8_t *p_, *end_;
    std::string what_;
};

// the members of the object whose SEALHeader starts at the reader's position; `owned` holds them when they had to be
// decompressed, otherwise the returned Reader aliases the input
Reader open_object(Reader &in, std::vector<uint8_t> &owned);

class Writer {
  public:
    std::vector<uint8_t> buf;
    template <class T>
    void put(const T &v)
    {
        put_bytes(&v, sizeof(T));
    }
    void put_bytes(const void *p, size_t n);
    // nested object with compr_mode none: header + members
    void put_header(uint64_t members_size);
    void put_modulus(uint64_t value);
    void put_dynarray(const uint64_t *data, uint64_t count);
};

// header + (compressed) members -> file
void write_object_file(const std::string &path, const std::vector<uint8_t> &members, Compr mode);
std::vector<uint8_t> read_file(const std::string &path);
Compr compr_from_env(); // option seal_compr (options.hpp) = 0 none (default) | 1 zlib | 2 zstd
bool zstd_available();

struct Params {
    uint8_t scheme = kSchemeCkks;
    uint64_t N = 0;
    std::vector<uint64_t> primes;
    uint64_t plain_modulus = 0;
};
void put_params(Writer &w, const Params &p);
Params get_params(Reader &members);

struct CtHeader { // Ciphertext members before the data array
    ParmsId id{};
    bool is_ntt = true;
    uint64_t size = 2, N = 0, limbs = 0, correction_factor = 1;
    double scale = 1.0;
};
void put_ciphertext(Writer &w, const CtHeader &h, const uint64_t *data);
// reads the members of a Ciphertext; `data` points into the reader's storage (size * limbs * N words).  The array sits behind a 1-byte
// field, so the pointer is NOT 8-byte aligned: copy from it (memcpy / hipMemcpy, what every caller does), never dereference it as uint64_t
CtHeader get_ciphertext(Reader &members, const uint64_t *&data);

struct PtHeader {
    ParmsId id{};
    uint64_t coeff_count = 0;
    double scale = 1.0;
};
void put_plaintext(Writer &w, const PtHeader &h, const uint64_t *data);
PtHeader get_plaintext(Reader &members, const uint64_t *&data);

} // namespace sealio
} // namespace dacapo
