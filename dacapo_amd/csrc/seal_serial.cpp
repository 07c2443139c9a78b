// SEAL 4.0 serialization (see seal_serial.hpp for the format and the reference call sites it serves).  Host code only.
#include "seal_serial.hpp"

#include <algorithm>
#include <stdexcept>

#include "options.hpp"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

namespace dacapo {
namespace sealio {

// ---- BLAKE2b (RFC 7693) ---------------------------------------------------------------------------------------------------
static const uint64_t kIv[8] = { 0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull };
static const uint8_t kSigma[12][16] = {
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
    { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
    { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
    { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
    { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 },
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 }
};
static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

static void b2_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last)
{
    uint64_t m[16], v[16];
    memcpy(m, block, 128); // little-endian host
    for (int i = 0; i < 8; i++) v[i] = h[i], v[i + 8] = kIv[i];
    v[12] ^= t; // messages here are far below 2^64 bytes: the high counter word stays 0
    if (last) v[14] = ~v[14];
    for (int r = 0; r < 12; r++) {
        const uint8_t *s = kSigma[r];
#define B2G(a, b, c, d, x, y)                                                                                                  \
    v[a] = v[a] + v[b] + (x), v[d] = rotr64(v[d] ^ v[a], 32), v[c] = v[c] + v[d], v[b] = rotr64(v[b] ^ v[c], 24),                   \
    v[a] = v[a] + v[b] + (y), v[d] = rotr64(v[d] ^ v[a], 16), v[c] = v[c] + v[d], v[b] = rotr64(v[b] ^ v[c], 63)
        B2G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        B2G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        B2G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        B2G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        B2G(3, 4, 9, 14, m[s[14]], m[s[15]]);
#undef B2G
    }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

void blake2b(void *out, size_t outlen, const void *in, size_t inlen)
{
    if (outlen == 0 || outlen > 64) {
        fprintf(stderr, "[dacapo_amd] blake2b: digest length %zu outside 1..64\n", outlen);
        abort();
    }
    uint64_t h[8];
    for (int i = 0; i < 8; i++) h[i] = kIv[i];
    h[0] ^= 0x01010000ull ^ (uint64_t)outlen;
    const uint8_t *p = (const uint8_t *)in;
    uint64_t t = 0;
    while (inlen > 128) { // the final block (possibly full) is compressed with the last flag
        t += 128;
        b2_compress(h, p, t, false);
        p += 128, inlen -= 128;
    }
    uint8_t block[128] = { 0 };
    memcpy(block, p, inlen);
    t += inlen;
    b2_compress(h, block, t, true);
    memcpy(out, h, outlen);
}

ParmsId parms_id(uint64_t N, const uint64_t *primes, size_t count, uint8_t scheme, uint64_t plain_modulus)
{ // EncryptionParameters::compute_parms_id
    std::vector<uint64_t> words;
    words.push_back(scheme);
    words.push_back(N);
    words.insert(words.end(), primes, primes + count);
    words.push_back(plain_modulus);
    ParmsId id;
    blake2b(id.data(), 32, words.data(), words.size() * 8);
    return id;
}

// ---- reader / writer ---------------------------------------------------------------------------------------------------------
void Reader::fail(const char *msg) const
{
#ifdef DC_PARSE_THROWS // the sanitizer harness (host_fuzz_main.cpp): a rejected input is an outcome there, not the end of the process
    throw std::runtime_error(what_ + ": " + msg);
#else
    fprintf(stderr, "[dacapo_amd] SEAL serialization: %s: %s\n", what_.c_str(), msg);
    abort();
#endif
}
void Reader::take(void *dst, size_t n)
{
    if (left() < n) fail("truncated");
    memcpy(dst, p_, n);
    p_ += n;
}
const uint8_t *Reader::skip(size_t n)
{
    if (left() < n) fail("truncated");
    const uint8_t *r = p_;
    p_ += n;
    return r;
}

// What a compressed object may expand to.  Everything SEAL stores in these files is residues mod ~60-bit primes in 64-bit words -- it
// deflates by a few per cent -- plus headers; the one legitimate exception is a transparent / all-zero ciphertext, which deflates ~1000 : 1.
// The largest object this library supports is a ciphertext of N = 2^17 at 40 primes: 2 x 40 x 2^17 x 8 = 80 MiB (round-5 advisor: the 16 MiB
// slack argued for the reference's ring refused an all-zero N = 2^17 ciphertext).  A stream that expands beyond 64 x its stored size + 128 MiB is
// not a key file, and inflating it on trust is how a few KB of input take down the host (zlib reaches 1032 : 1, Zstandard far more).
static size_t inflate_limit(size_t stored) { return stored * 64 + ((size_t)128 << 20); }
static const char *const kBombMsg = "compressed members expand beyond 64 x their stored size + 128 MiB: refusing (not a SEAL key / ciphertext object; decompression bomb?)";

static std::vector<uint8_t> inflate_all(const uint8_t *in, size_t n, const Reader &r)
{
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) r.fail("inflateInit failed");
    const size_t limit = inflate_limit(n);
    std::vector<uint8_t> out(n * 2 + (1 << 16));
    size_t in_pos = 0, out_pos = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        if (zs.avail_in == 0 && in_pos < n) {
            const size_t chunk = n - in_pos < (1u << 30) ? n - in_pos : (1u << 30);
            zs.next_in = const_cast<Bytef *>(in + in_pos), zs.avail_in = (uInt)chunk, in_pos += chunk;
        }
        if (out_pos == out.size()) {
            if (out.size() >= limit) {
                inflateEnd(&zs);
                r.fail(kBombMsg);
            }
            out.resize(std::min(out.size() * 2, limit));
        }
        const size_t room = out.size() - out_pos < (1u << 30) ? out.size() - out_pos : (1u << 30);
        zs.next_out = out.data() + out_pos, zs.avail_out = (uInt)room;
        rc = inflate(&zs, Z_NO_FLUSH);
        out_pos += room - zs.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
            inflateEnd(&zs);
            r.fail("zlib stream is corrupt");
        }
        if (rc == Z_BUF_ERROR && zs.avail_in == 0 && in_pos >= n) {
            inflateEnd(&zs);
            r.fail("zlib stream is truncated");
        }
    }
    inflateEnd(&zs);
    out.resize(out_pos);
    return out;
}

static std::vector<uint8_t> deflate_all(const uint8_t *in, size_t n)
{
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit(&zs, Z_DEFAULT_COMPRESSION) != Z_OK) {
        fprintf(stderr, "[dacapo_amd] SEAL serialization: deflateInit failed\n");
        abort();
    }
    std::vector<uint8_t> out(n + n / 1000 + (1 << 16));
    size_t in_pos = 0, out_pos = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        if (zs.avail_in == 0 && in_pos < n) {
            const size_t chunk = n - in_pos < (1u << 30) ? n - in_pos : (1u << 30);
            zs.next_in = const_cast<Bytef *>(in + in_pos), zs.avail_in = (uInt)chunk, in_pos += chunk;
        }
        if (out_pos == out.size()) out.resize(out.size() * 2);
        const size_t room = out.size() - out_pos < (1u << 30) ? out.size() - out_pos : (1u << 30);
        zs.next_out = out.data() + out_pos, zs.avail_out = (uInt)room;
        rc = deflate(&zs, in_pos >= n ? Z_FINISH : Z_NO_FLUSH);
        out_pos += room - zs.avail_out;
        if (rc == Z_STREAM_ERROR) {
            fprintf(stderr, "[dacapo_amd] SEAL serialization: deflate failed\n");
            abort();
        }
    }
    deflateEnd(&zs);
    out.resize(out_pos);
    return out;
}

// Zstandard: the image ships libzstd.so.1 without its header, so the handful of stable public entry points used here are
// declared locally (zstd.h, "Simple API" and "Streaming decompression") and resolved at run time.
struct ZstdApi {
    struct InBuf { const void *src; size_t size, pos; };
    struct OutBuf { void *dst; size_t size, pos; };
    void *(*createDStream)() = nullptr;
    size_t (*freeDStream)(void *) = nullptr;
    size_t (*decompressStream)(void *, OutBuf *, InBuf *) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    size_t (*compressBound)(size_t) = nullptr;
    size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
    bool ok = false;
    ZstdApi()
    {
        void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        createDStream = (void *(*)())dlsym(h, "ZSTD_createDStream");
        freeDStream = (size_t(*)(void *))dlsym(h, "ZSTD_freeDStream");
        decompressStream = (size_t(*)(void *, OutBuf *, InBuf *))dlsym(h, "ZSTD_decompressStream");
        isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        compressBound = (size_t(*)(size_t))dlsym(h, "ZSTD_compressBound");
        compress = (size_t(*)(void *, size_t, const void *, size_t, int))dlsym(h, "ZSTD_compress");
        ok = createDStream && freeDStream && decompressStream && isError && compressBound && compress;
    }
};
static const ZstdApi &zstd()
{
    static const ZstdApi api;
    return api;
}
bool zstd_available() { return zstd().ok; }

static std::vector<uint8_t> zstd_inflate_all(const uint8_t *in, size_t n, const Reader &r)
{
    const ZstdApi &z = zstd();
    if (!z.ok) r.fail("object is Zstandard-compressed and libzstd.so.1 is not available; re-save it with compr_mode_type::zlib or ::none");
    void *ds = z.createDStream();
    if (!ds) r.fail("ZSTD_createDStream failed");
    const size_t limit = inflate_limit(n);
    std::vector<uint8_t> out(n * 2 + (1 << 16));
    ZstdApi::InBuf ib{ in, n, 0 };
    size_t out_pos = 0, rc = 1;
    while (ib.pos < ib.size || rc != 0) { // SEAL writes one frame; rc == 0 marks its end
        if (out_pos == out.size()) {
            if (out.size() >= limit) {
                z.freeDStream(ds);
                r.fail(kBombMsg);
            }
            out.resize(std::min(out.size() * 2, limit));
        }
        ZstdApi::OutBuf ob{ out.data() + out_pos, out.size() - out_pos, 0 };
        rc = z.decompressStream(ds, &ob, &ib);
        if (z.isError(rc)) {
            z.freeDStream(ds);
            r.fail("Zstandard stream is corrupt");
        }
        out_pos += ob.pos;
        if (rc != 0 && ib.pos == ib.size && ob.pos == 0) {
            z.freeDStream(ds);
            r.fail("Zstandard stream is truncated");
        }
        if (rc == 0 && ib.pos == ib.size) break;
    }
    z.freeDStream(ds);
    out.resize(out_pos);
    return out;
}

#pragma pack(push, 1)
struct SealHeader {
    uint16_t magic;
    uint8_t header_size, version_major, version_minor, compr_mode;
    uint16_t reserved;
    uint64_t size;
};
#pragma pack(pop)
static_assert(sizeof(SealHeader) == 16, "SEALHeader is 16 bytes");

Reader open_object(Reader &in, std::vector<uint8_t> &owned)
{
    const SealHeader h = in.get<SealHeader>();
    if (h.magic != kMagic || h.header_size != 16) in.fail("not a SEAL-serialized object (bad magic / header size)");
    // 3.6 / 3.7 headers look the same, but their Ciphertext layout has no correction_factor field and every key object here is made of
    // ciphertexts: accepting them would misparse pub / relin / gal files with a misleading "array length" error.  The reference pins 4.0.
    if (h.version_major != 4)
        in.fail("written by SEAL 3.x or another unsupported version: this runtime reads SEAL 4.x serialization only (re-save the keys with SEAL 4)");
    if (h.size < 16 || h.size - 16 > in.left()) in.fail("header size field exceeds the data");
    const size_t stored = (size_t)(h.size - 16);
    const uint8_t *body = in.skip(stored);
    const std::string what = in.what();
    switch (h.compr_mode) {
    case COMPR_NONE: return Reader(body, stored, what);
    case COMPR_ZLIB: owned = inflate_all(body, stored, in); break;
    case COMPR_ZSTD: owned = zstd_inflate_all(body, stored, in); break;
    default: in.fail("unknown compr_mode");
    }
    return Reader(owned.data(), owned.size(), what);
}

void Writer::put_bytes(const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    buf.insert(buf.end(), b, b + n);
}
void Writer::put_header(uint64_t members_size)
{
    const SealHeader h{ kMagic, 16, 4, 0, COMPR_NONE, 0, 16 + members_size };
    put(h);
}
void Writer::put_modulus(uint64_t value)
{
    put_header(8);
    put(value);
}
void Writer::put_dynarray(const uint64_t *data, uint64_t count)
{
    put_header(8 + 8 * count);
    put(count);
    if (count) put_bytes(data, (size_t)count * 8);
}

Compr compr_from_env()
{
    const long long v = dacapo::option(dacapo::OPT_SEAL_COMPR);
    if (v == 0) return COMPR_NONE;
    if (v == 1) return COMPR_ZLIB;
    if (v == 2) return COMPR_ZSTD;
    fprintf(stderr, "[dacapo_amd] option seal_compr = %lld: expected 0 (none), 1 (zlib) or 2 (zstd)\n", v);
    abort();
}

void write_object_file(const std::string &path, const std::vector<uint8_t> &members, Compr mode)
{
    std::vector<uint8_t> packed;
    const std::vector<uint8_t> *body = &members;
    if (mode == COMPR_ZLIB) {
        packed = deflate_all(members.data(), members.size());
        body = &packed;
    } else if (mode == COMPR_ZSTD) {
        const ZstdApi &z = zstd();
        if (!z.ok) {
            fprintf(stderr, "[dacapo_amd] SEAL serialization: zstd requested but libzstd.so.1 is not available\n");
            abort();
        }
        packed.resize(z.compressBound(members.size()));
        const size_t n = z.compress(packed.data(), packed.size(), members.data(), members.size(), 3);
        if (z.isError(n)) {
            fprintf(stderr, "[dacapo_amd] SEAL serialization: ZSTD_compress failed\n");
            abort();
        }
        packed.resize(n);
        body = &packed;
    }
    const SealHeader h{ kMagic, 16, 4, 0, (uint8_t)mode, 0, 16 + (uint64_t)body->size() };
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(&h, 16, 1, f) != 1 || (body->size() && fwrite(body->data(), 1, body->size(), f) != body->size()) || fclose(f) != 0) {
        fprintf(stderr, "[dacapo_amd] cannot write %s\n", path.c_str());
        abort();
    }
}

std::vector<uint8_t> read_file(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "[dacapo_amd] cannot open %s\n", path.c_str());
        abort();
    }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)(n > 0 ? n : 0));
    if (n > 0 && fread(buf.data(), 1, (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "[dacapo_amd] cannot read %s\n", path.c_str());
        abort();
    }
    fclose(f);
    return buf;
}

// ---- objects -------------------------------------------------------------------------------------------------------------------
void put_params(Writer &w, const Params &p)
{ // EncryptionParameters::save_members
    w.put<uint8_t>(p.scheme);
    w.put<uint64_t>(p.N);
    w.put<uint64_t>(p.primes.size());
    for (uint64_t q : p.primes) w.put_modulus(q);
    w.put_modulus(p.plain_modulus);
}

static uint64_t get_modulus(Reader &r)
{
    std::vector<uint8_t> owned;
    Reader m = open_object(r, owned);
    return m.get<uint64_t>();
}

Params get_params(Reader &r)
{ // EncryptionParameters::load_members
    Params p;
    p.scheme = r.get<uint8_t>();
    p.N = r.get<uint64_t>();
    const uint64_t k = r.get<uint64_t>();
    if (p.N < 2 || p.N > (1ull << 20) || (p.N & (p.N - 1)) || k < 1 || k > 64) r.fail("implausible poly_modulus_degree / coeff_modulus_size");
    for (uint64_t i = 0; i < k; i++) p.primes.push_back(get_modulus(r));
    p.plain_modulus = get_modulus(r);
    return p;
}

static const uint64_t *get_dynarray(Reader &r, uint64_t &count, std::vector<uint8_t> &owned)
{
    Reader m = open_object(r, owned);
    count = m.get<uint64_t>();
    if (count > m.left() / 8) m.fail("DynArray count exceeds its data");
    return reinterpret_cast<const uint64_t *>(m.skip((size_t)count * 8));
}

void put_ciphertext(Writer &w, const CtHeader &h, const uint64_t *data)
{ // Ciphertext::save_members
    w.put(h.id);
    w.put<uint8_t>(h.is_ntt ? 1 : 0);
    w.put<uint64_t>(h.size);
    w.put<uint64_t>(h.N);
    w.put<uint64_t>(h.limbs);
    w.put<uint64_t>(h.correction_factor);
    w.put<double>(h.scale);
    w.put_dynarray(data, h.size * h.N * h.limbs);
}

CtHeader get_ciphertext(Reader &r, const uint64_t *&data)
{ // Ciphertext::load_members
    CtHeader h;
    h.id = r.get<ParmsId>();
    h.is_ntt = r.get<uint8_t>() != 0;
    h.size = r.get<uint64_t>();
    h.N = r.get<uint64_t>();
    h.limbs = r.get<uint64_t>();
    h.correction_factor = r.get<uint64_t>();
    h.scale = r.get<double>();
    if (h.size > 16 || h.limbs > 64 || h.N > (1ull << 20)) r.fail("implausible ciphertext dimensions");
    static thread_local std::vector<uint8_t> owned; // nested arrays are stored uncompressed by SEAL; keep a decompressed one alive
    uint64_t count = 0;
    data = get_dynarray(r, count, owned);
    if (h.size >= 2 && count == h.N * h.limbs)
        r.fail("seed-compressed object (saved through Serializable<>): expanding SEAL's Blake2xb/SHAKE256 seed is not supported; "
               "save the full object as the reference does (SEAL_HEVM.cpp:62-86)");
    if (count != h.size * h.N * h.limbs) r.fail("data array length does not match size * coeff_modulus_size * poly_modulus_degree");
    return h;
}

void put_plaintext(Writer &w, const PtHeader &h, const uint64_t *data)
{ // Plaintext::save_members
    w.put(h.id);
    w.put<uint64_t>(h.coeff_count);
    w.put<double>(h.scale);
    w.put_dynarray(data, h.coeff_count);
}

PtHeader get_plaintext(Reader &r, const uint64_t *&data)
{ // Plaintext::load_members
    PtHeader h;
    h.id = r.get<ParmsId>();
    h.coeff_count = r.get<uint64_t>();
    h.scale = r.get<double>();
    static thread_local std::vector<uint8_t> owned;
    uint64_t count = 0;
    data = get_dynarray(r, count, owned);
    if (count != h.coeff_count) r.fail("data array length does not match coeff_count");
    return h;
}

} // namespace sealio
} // namespace dacapo
