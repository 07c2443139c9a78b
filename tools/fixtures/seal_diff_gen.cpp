// Differential-test generator: run Microsoft SEAL itself (4.0, the library the reference links, README.md:65-73) on a fixed
// little scenario and save everything in SEAL's own serialization, so that tests/test_seal_diff.py can feed the same keys
// and ciphertexts to this repo's runtime (initFullVM / hevm_load_ctxt) and to the oracle, and compare result limbs.
// Not built by default: SEAL is not installed in the build image.  On a machine that has it:
//     g++ -std=c++17 -O2 tools/fixtures/seal_diff_gen.cpp -I$SEAL/include/SEAL-4.0 -L$SEAL/lib -lseal-4.0 -o seal_diff_gen
//     ./seal_diff_gen <outdir> [logN=13] [primes=5]
// The key directory is written exactly as SEAL_HEVM::create_context writes it (SEAL_HEVM.cpp:44-89: parm / pub / sec /
// relin / gal .seal, default compr_mode, full -- not seed-compressed -- keys, default Galois key set); the evaluator calls
// are those of the opcode handlers (SEAL_HEVM.cpp:268-323).
#include <seal/seal.h>

#include <cmath>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

using namespace seal;

template <class T>
static void save_to(const T &obj, const std::string &path)
{
    std::ofstream f(path, std::ios::out | std::ios::binary);
    obj.save(f);
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::cerr << "usage: seal_diff_gen <outdir> [logN] [primes]" << std::endl;
        return 2;
    }
    const std::string dir = argv[1];
    const int logN = argc > 2 ? std::atoi(argv[2]) : 13;
    const int K = argc > 3 ? std::atoi(argv[3]) : 5;
    const size_t N = size_t(1) << logN;

    EncryptionParameters parms(scheme_type::ckks);
    parms.set_poly_modulus_degree(N);
    parms.set_coeff_modulus(CoeffModulus::Create(N, std::vector<int>(static_cast<size_t>(K), 60)));
    save_to(parms, dir + "/parm.seal");
    SEALContext context(parms, true, sec_level_type::none); // small rings for a quick test: no security-level check
    KeyGenerator keygen(context);
    PublicKey pubkey;
    keygen.create_public_key(pubkey);
    save_to(pubkey, dir + "/pub.seal");
    SecretKey seckey = keygen.secret_key();
    save_to(seckey, dir + "/sec.seal");
    RelinKeys relkey;
    keygen.create_relin_keys(relkey);
    save_to(relkey, dir + "/relin.seal");
    GaloisKeys galkey;
    keygen.create_galois_keys(galkey);
    save_to(galkey, dir + "/gal.seal");

    Encryptor encryptor(context, pubkey);
    Evaluator evaluator(context);
    CKKSEncoder encoder(context);
    const size_t slots = N / 2;
    std::vector<double> a(slots), b(slots);
    for (size_t i = 0; i < slots; i++) {
        a[i] = std::sin(0.001 * static_cast<double>(i));
        b[i] = 0.5 * std::cos(0.002 * static_cast<double>(i));
    }
    std::ofstream va(dir + "/a.f64", std::ios::binary), vb(dir + "/b.f64", std::ios::binary);
    va.write(reinterpret_cast<const char *>(a.data()), static_cast<std::streamsize>(slots * 8));
    vb.write(reinterpret_cast<const char *>(b.data()), static_cast<std::streamsize>(slots * 8));
    const double scale = std::pow(2.0, 40);
    Plaintext pa, pb;
    encoder.encode(a, scale, pa); // top data level, like SEAL_HEVM::encode_internal before its mod-switch loop (:262)
    encoder.encode(b, scale, pb);
    save_to(pa, dir + "/a.pt");
    Ciphertext ca, cb, r;
    encryptor.encrypt(pa, ca);
    encryptor.encrypt(pb, cb);
    save_to(ca, dir + "/a.ct");
    save_to(cb, dir + "/b.ct");

    for (int step : { 1, 37, -100 }) { // a direct key, a 3-hop and a 3-hop negative NAF decomposition (SEAL_HEVM.cpp:273)
        evaluator.rotate_vector(ca, step, galkey, r);
        save_to(r, dir + "/rotate_" + std::to_string(step) + ".ct");
    }
    evaluator.negate(ca, r); // :278
    save_to(r, dir + "/negate.ct");
    evaluator.add(ca, cb, r); // :302
    save_to(r, dir + "/add.ct");
    evaluator.mod_switch_to_next(ca, r); // :289
    save_to(r, dir + "/modswitch.ct");
    Ciphertext m;
    evaluator.multiply(ca, cb, m); // :315
    evaluator.relinearize_inplace(m, relkey); // :316
    save_to(m, dir + "/mul.ct");
    evaluator.rescale_to_next(m, r); // :283
    save_to(r, dir + "/rescale.ct");
    evaluator.multiply_plain(ca, pb, r); // :322
    save_to(r, dir + "/mulcp.ct");
    evaluator.add_plain(ca, pb, r); // :309
    save_to(r, dir + "/addcp.ct");

    Decryptor decryptor(context, seckey);
    Plaintext out;
    decryptor.decrypt(m, out);
    std::vector<double> dec;
    encoder.decode(out, dec);
    std::ofstream vd(dir + "/mul.decoded.f64", std::ios::binary);
    vd.write(reinterpret_cast<const char *>(dec.data()), static_cast<std::streamsize>(dec.size() * 8));
    std::cout << "wrote SEAL " << SEAL_VERSION_MAJOR << "." << SEAL_VERSION_MINOR << " scenario to " << dir << std::endl;
    return 0;
}
