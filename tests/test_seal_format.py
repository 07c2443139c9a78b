"""CPU: SEAL 4.0 serialization and the key generator's PRNG -- host code of the product (libSEAL_HEVM.so, no GPU call) against
independent restatements: oracle/seal_format.py (struct / zlib / hashlib), a byte-by-byte hand-assembled object, hashlib's
BLAKE2b, and the RFC 8439 ChaCha20 vectors.  The reference side: SEAL_HEVM.cpp:55-88 writes, :91-180 reads these objects."""
import ctypes as C
import hashlib
import struct

import numpy as np
import pytest

from dacapo_amd import LIB_PATH
from oracle import seal_format as sf

PRIMES_15 = [0xffffffffe7c0001, 0xffffffffe830001, 0xffffffffe9e0001, 0xffffffffebb0001, 0xffffffffeca0001, 0xffffffffefe0001,
             0xfffffffff240001, 0xfffffffff2a0001, 0xfffffffff330001, 0xfffffffff550001, 0xfffffffff5a0001, 0xfffffffff6a0001,
             0xfffffffff840001, 0xffffffffffc0001]  # CoeffModulus::Create(2^15, {60 x 14})


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(str(LIB_PATH))
    u64p, i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_int)
    L.hevm_seal_parms_id.argtypes = [C.c_uint64, u64p, C.c_int, u64p]
    L.hevm_seal_save_parms.argtypes = [C.c_char_p, C.c_int, C.c_uint64, u64p, C.c_int]
    L.hevm_seal_load_parms.argtypes = [C.c_char_p, u64p, u64p, C.c_int]
    L.hevm_seal_save_ciphertext.argtypes = [C.c_char_p, C.c_int, C.c_uint64, u64p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    L.hevm_seal_load_ciphertext.restype = C.c_int64
    L.hevm_seal_load_ciphertext.argtypes = [C.c_char_p, u64p, i32p, i32p, i32p, C.POINTER(C.c_double), u64p, C.c_void_p, C.c_uint64]
    L.hevm_chacha20_block.argtypes = [C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    return L


def u64arr(xs):
    return (C.c_uint64 * len(xs))(*[int(x) for x in xs])


def lib_parms_id(lib, N, primes):
    out = (C.c_uint64 * 4)()
    lib.hevm_seal_parms_id(N, u64arr(primes), len(primes), out)
    return tuple(out)


def test_parms_id_is_blake2b_256_of_the_parameter_words(lib):
    for N, primes in ((1 << 15, PRIMES_15), (1 << 15, PRIMES_15[:13]), (1 << 15, PRIMES_15[:1]), (8, [97]), (1 << 12, PRIMES_15[:20 - 14])):
        words = [2, N] + primes + [0]  # scheme ckks, degree, coefficient moduli, plain modulus
        want = struct.unpack("<4Q", hashlib.blake2b(struct.pack(f"<{len(words)}Q", *words), digest_size=32).digest())
        assert lib_parms_id(lib, N, primes) == want == sf.parms_id(N, primes)


def test_blake2b_message_lengths_around_the_block_size(lib):
    # 14, 15, 16, 17 words = 112 .. 136 bytes straddle BLAKE2b's 128-byte block (the full-block-is-last case included)
    for count in (11, 12, 13, 14, 15, 29, 30, 31):
        primes = [(0xfffffffffffc0001 - 0x20001 * i) & (2**64 - 1) for i in range(count)]
        words = [2, 64] + primes + [0]
        want = struct.unpack("<4Q", hashlib.blake2b(struct.pack(f"<{len(words)}Q", *words), digest_size=32).digest())
        assert lib_parms_id(lib, 64, primes) == want


def test_parameters_file_byte_for_byte(lib, tmp_path):
    """EncryptionParameters::save with compr_mode none, assembled by hand from the documented layout"""
    N, primes = 8, [0xffffffffffc0001, 0xfffffffff840001]
    p = tmp_path / "parm.seal"
    lib.hevm_seal_save_parms(str(p).encode(), 0, N, u64arr(primes), len(primes))
    hdr = lambda total: struct.pack("<HBBBBHQ", 0xA15E, 16, 4, 0, 0, 0, total)  # noqa: E731
    mod = lambda q: hdr(24) + struct.pack("<Q", q)  # noqa: E731
    members = struct.pack("<B", 2) + struct.pack("<Q", N) + struct.pack("<Q", 2) + mod(primes[0]) + mod(primes[1]) + mod(0)
    assert p.read_bytes() == hdr(16 + len(members)) + members
    assert len(p.read_bytes()) == 16 + 17 + 3 * 24
    # and the Python writer agrees
    assert sf.wrap(sf.params_members(N, primes)) == p.read_bytes()


@pytest.mark.parametrize("compr", [0, 1, 2])
def test_parameters_round_trip_both_directions(lib, tmp_path, compr):
    if compr == 2 and not lib.hevm_seal_zstd_available():
        pytest.skip("libzstd.so.1 not present")
    p = tmp_path / "parm.seal"
    lib.hevm_seal_save_parms(str(p).encode(), compr, 1 << 15, u64arr(PRIMES_15), 14)
    n, out = C.c_uint64(), (C.c_uint64 * 32)()
    assert lib.hevm_seal_load_parms(str(p).encode(), C.byref(n), out, 32) == 14
    assert n.value == 1 << 15 and list(out)[:14] == PRIMES_15
    if compr != 2:  # product wrote, Python reads ...
        got = sf.read_params_members(sf.unwrap(p.read_bytes())[0])
        assert got == {"scheme": 2, "N": 1 << 15, "primes": PRIMES_15, "plain_modulus": 0}
        # ... Python writes, product reads
        p.write_bytes(sf.wrap(sf.params_members(1 << 12, PRIMES_15[:5]), compr))
        assert lib.hevm_seal_load_parms(str(p).encode(), C.byref(n), out, 32) == 5
        assert n.value == 1 << 12 and list(out)[:5] == PRIMES_15[:5]


@pytest.mark.parametrize("compr", [0, 1, 2])
def test_ciphertext_round_trip_both_directions(lib, tmp_path, compr):
    if compr == 2 and not lib.hevm_seal_zstd_available():
        pytest.skip("libzstd.so.1 not present")
    N, limbs = 1 << 10, 3
    rng = np.random.default_rng(5)
    data = (rng.integers(0, 1 << 62, size=(2, limbs, N), dtype=np.uint64) % np.array(PRIMES_15[:limbs], dtype=np.uint64)[None, :, None])
    p = tmp_path / "ct.seal"
    lib.hevm_seal_save_ciphertext(str(p).encode(), compr, N, u64arr(PRIMES_15), limbs, 2, 1, 2.0**40, data.ctypes.data)

    def load():
        n, l, s, ntt, sc = C.c_uint64(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
        pid, buf = (C.c_uint64 * 4)(), np.zeros(2 * limbs * N, dtype=np.uint64)
        words = lib.hevm_seal_load_ciphertext(str(p).encode(), C.byref(n), C.byref(l), C.byref(s), C.byref(ntt), C.byref(sc), pid,
                                              buf.ctypes.data, buf.size)
        return words, n.value, l.value, s.value, ntt.value, sc.value, tuple(pid), buf.reshape(2, limbs, N)

    words, n, l, s, ntt, sc, pid, buf = load()
    assert (words, n, l, s, ntt, sc) == (2 * limbs * N, N, limbs, 2, 1, 2.0**40)
    assert pid == sf.parms_id(N, PRIMES_15[:limbs]) and (buf == data).all()  # the id of the chain truncated to the ciphertext's level
    if compr != 2:
        got, _ = sf.read_ciphertext_members(sf.unwrap(p.read_bytes())[0])
        assert got["parms_id"] == pid and got["scale"] == 2.0**40 and got["is_ntt"] and got["correction_factor"] == 1
        assert (got["data"] == data).all()
        p.write_bytes(sf.wrap(sf.ciphertext_members(pid, data[:, ::-1].copy(), scale=3.5), compr))
        words, n, l, s, ntt, sc, pid2, buf = load()
        assert sc == 3.5 and pid2 == pid and (buf == data[:, ::-1]).all()


def test_ciphertext_layout_byte_for_byte(lib, tmp_path):
    """Ciphertext::save_members: parms_id, is_ntt_form, size, degree, limb count, correction factor, scale, DynArray"""
    N, primes = 8, [0xffffffffffc0001]
    data = np.arange(16, dtype=np.uint64).reshape(2, 1, 8)
    p = tmp_path / "ct.seal"
    lib.hevm_seal_save_ciphertext(str(p).encode(), 0, N, u64arr(primes), 1, 2, 1, 1.0, data.ctypes.data)
    hdr = lambda total: struct.pack("<HBBBBHQ", 0xA15E, 16, 4, 0, 0, 0, total)  # noqa: E731
    pid = hashlib.blake2b(struct.pack("<4Q", 2, 8, primes[0], 0), digest_size=32).digest()
    arr = hdr(16 + 8 + 128) + struct.pack("<Q", 16) + data.tobytes()
    members = pid + b"\x01" + struct.pack("<QQQQd", 2, 8, 1, 1, 1.0) + arr
    assert p.read_bytes() == hdr(16 + len(members)) + members


def test_foreign_and_seeded_objects_abort_with_a_message(tmp_path):
    import subprocess
    import sys

    bad = tmp_path / "bad.seal"
    bad.write_bytes(b"DCHEVM01" + bytes(64))
    code = ("import ctypes as C; L = C.CDLL(%r); n = C.c_uint64(); o = (C.c_uint64 * 4)();"
            "L.hevm_seal_load_parms(%r, C.byref(n), o, 4)" % (str(LIB_PATH), str(bad).encode()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "not a SEAL-serialized object" in r.stderr
    # a seed-compressed ciphertext (Serializable<Ciphertext>): only c0 is stored
    N = 16
    half = np.zeros((1, 1, N), dtype=np.uint64)
    m = struct.pack("<4Q", *sf.parms_id(N, PRIMES_15[:1])) + struct.pack("<BQQQQd", 1, 2, N, 1, 1, 1.0) + sf._dynarray(half)
    bad.write_bytes(sf.wrap(m))
    code = ("import ctypes as C; L = C.CDLL(%r); L.hevm_seal_load_ciphertext.restype = C.c_int64;"
            "n = C.c_uint64(); a = C.c_int(); b = C.c_int(); c = C.c_int(); d = C.c_double(); o = (C.c_uint64 * 4)();"
            "L.hevm_seal_load_ciphertext(%r, C.byref(n), C.byref(a), C.byref(b), C.byref(c), C.byref(d), o, None, C.c_uint64(0))"
            % (str(LIB_PATH), str(bad).encode()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "seed-compressed" in r.stderr


def test_key_directory_python_round_trip(tmp_path):
    """oracle-side writer and reader agree on the five files (RelinKeys dim1 = 1, GaloisKeys dim1 = N sparse)"""
    N, K = 32, 3
    rng = np.random.default_rng(1)
    primes = PRIMES_15[:K]
    pk = rng.integers(0, 1 << 59, size=(2, K, N), dtype=np.uint64)
    sk = rng.integers(0, 1 << 59, size=(K, N), dtype=np.uint64)
    relin = rng.integers(0, 1 << 59, size=(K - 1, 2, K, N), dtype=np.uint64)
    gal = {3: relin + np.uint64(1), 2 * N - 1: relin + np.uint64(2), 9: relin + np.uint64(3)}
    sf.write_key_dir(tmp_path, N, primes, pk, sk, relin, gal, compr=sf.COMPR_ZLIB)
    got = sf.read_key_dir(tmp_path)
    kid = sf.parms_id(N, primes)
    assert got["params"]["primes"] == primes and got["pk"]["parms_id"] == kid and (got["pk"]["data"] == pk).all()
    assert (got["sk"]["data"].reshape(K, N) == sk).all() and got["sk"]["scale"] == 1.0
    assert got["relin"]["dim1"] == 1 and (got["relin"]["present"][0] == relin).all()
    assert got["gal"]["dim1"] == N and sorted(got["gal"]["by_elt"]) == [3, 9, 2 * N - 1]
    assert all((got["gal"]["by_elt"][e] == gal[e]).all() for e in gal)


# ---- ChaCha20 ---------------------------------------------------------------------------------------------------------------------
def chacha20_block_py(key_words, counter, nonce):
    """RFC 8439 section 2.3 restated with Python integers; state words 12-13 = counter, 14-15 = nonce"""
    M = 0xFFFFFFFF
    rotl = lambda x, n: ((x << n) | (x >> (32 - n))) & M  # noqa: E731
    s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(key_words) + [counter & M, counter >> 32, nonce & M, nonce >> 32]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M for a, b in zip(x, s)]


def lib_block(lib, key_words, counter, nonce):
    key, out = (C.c_uint32 * 8)(*key_words), (C.c_uint32 * 16)()
    lib.hevm_chacha20_block(key, counter, nonce, out)
    return list(out)


def test_chacha20_known_answers(lib):
    # all-zero key, counter, nonce: the classic first keystream block 76 b8 e0 ad a0 f1 3d 90 ...
    zero = lib_block(lib, [0] * 8, 0, 0)
    assert struct.pack("<16I", *zero)[:16] == bytes.fromhex("76b8e0ada0f13d90405d6ae55386bd28")
    # RFC 8439 2.3.2: key 00..1f, block counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00.  In the 64/64 split used here the
    # RFC's 32-bit counter and first nonce word are the low/high halves of `counter`, the remaining nonce words are `nonce`.
    key = list(struct.unpack("<8I", bytes(range(32))))
    got = lib_block(lib, key, 1 | (0x09000000 << 32), 0x4a000000)
    assert got[:4] == [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3]
    assert got == chacha20_block_py(key, 1 | (0x09000000 << 32), 0x4a000000)


def test_chacha20_matches_the_python_restatement_on_random_inputs(lib):
    rng = np.random.default_rng(3)
    for _ in range(50):
        key = [int(x) for x in rng.integers(0, 1 << 32, size=8)]
        ctr, nonce = int(rng.integers(0, 1 << 63)) * 2 + 1, int(rng.integers(0, 1 << 63))
        assert lib_block(lib, key, ctr, nonce) == chacha20_block_py(key, ctr, nonce)
