"""TEST INFRASTRUCTURE (oracle side): Microsoft SEAL 4.0 binary serialization in plain Python -- struct, zlib, hashlib.

An implementation independent of dacapo_amd/csrc/seal_serial.cpp (the product's C++ reader/writer): the tests make each read
what the other wrote.  Only tests/ may import this module.  Format restated from SEAL 4.0.0, the library the reference
links (/root/reference/README.md:65-73) and whose save/load calls produce the reference's key directory
(/root/reference/lib/Runtime/SEAL_HEVM.cpp:55-88, :91-180) [SEAL-upstream native/src/seal/serialization.h,
encryptionparams.cpp, ciphertext.cpp, plaintext.cpp, kswitchkeys.cpp, dynarray.h; SEAL itself is not available here, so
"parity unpinned" applies to this restatement too until tests/test_seal_diff.py runs on a machine that has SEAL].
"""
from __future__ import annotations

import hashlib
import struct
import zlib
from pathlib import Path

import numpy as np

MAGIC, HEADER_SIZE, VERSION = 0xA15E, 16, (4, 0)
COMPR_NONE, COMPR_ZLIB, COMPR_ZSTD = 0, 1, 2
SCHEME_CKKS = 2


def header(total_size: int, compr: int = COMPR_NONE) -> bytes:
    """Serialization::SEALHeader: u16 magic, u8 header_size, u8 major, u8 minor, u8 compr_mode, u16 reserved, u64 size"""
    return struct.pack("<HBBBBHQ", MAGIC, HEADER_SIZE, VERSION[0], VERSION[1], compr, 0, total_size)


def wrap(members: bytes, compr: int = COMPR_NONE) -> bytes:
    if compr == COMPR_ZLIB:
        members = zlib.compress(members)
    elif compr != COMPR_NONE:
        raise ValueError("only none / zlib can be written from Python")
    return header(HEADER_SIZE + len(members), compr) + members


def unwrap(buf: bytes, off: int = 0):
    """-> (members, offset after the object)"""
    magic, hs, major, minor, compr, _res, size = struct.unpack_from("<HBBBBHQ", buf, off)
    if magic != MAGIC or hs != HEADER_SIZE:
        raise ValueError("not a SEAL object")
    body = bytes(buf[off + HEADER_SIZE: off + size])
    if len(body) != size - HEADER_SIZE:
        raise ValueError("truncated SEAL object")
    if compr == COMPR_ZLIB:
        body = zlib.decompress(body)
    elif compr != COMPR_NONE:
        raise ValueError(f"compr_mode {compr} not readable from Python")
    return body, off + size


def parms_id(N: int, primes, scheme: int = SCHEME_CKKS, plain_modulus: int = 0):
    """EncryptionParameters::compute_parms_id: BLAKE2b-256 over the u64 words {scheme, N, q_i..., plain_modulus}"""
    words = [scheme, N] + [int(q) for q in primes] + [plain_modulus]
    d = hashlib.blake2b(struct.pack(f"<{len(words)}Q", *words), digest_size=32).digest()
    return struct.unpack("<4Q", d)


def _dynarray(a: np.ndarray) -> bytes:
    a = np.ascontiguousarray(a, dtype="<u8").ravel()
    return wrap(struct.pack("<Q", a.size) + a.tobytes())


def _read_dynarray(buf: bytes, off: int):
    body, off = unwrap(buf, off)
    (n,) = struct.unpack_from("<Q", body, 0)
    return np.frombuffer(body, dtype="<u8", count=n, offset=8).copy(), off


def _modulus(q: int) -> bytes:
    return wrap(struct.pack("<Q", q))


# ---- EncryptionParameters -------------------------------------------------------------------------------------------------
def params_members(N: int, primes) -> bytes:
    out = struct.pack("<BQQ", SCHEME_CKKS, N, len(primes))
    for q in primes:
        out += _modulus(int(q))
    return out + _modulus(0)


def read_params_members(m: bytes):
    scheme, N, k = struct.unpack_from("<BQQ", m, 0)
    off, primes = 17, []
    for _ in range(k + 1):
        body, off = unwrap(m, off)
        primes.append(struct.unpack("<Q", body)[0])
    return {"scheme": scheme, "N": N, "primes": primes[:-1], "plain_modulus": primes[-1]}


# ---- Ciphertext / PublicKey ---------------------------------------------------------------------------------------------------
def ciphertext_members(pid, data: np.ndarray, scale: float = 1.0, is_ntt: bool = True, correction_factor: int = 1) -> bytes:
    """data: [size][limbs][N]"""
    size, limbs, N = data.shape
    return (struct.pack("<4Q", *pid) + struct.pack("<BQQQQd", int(is_ntt), size, N, limbs, correction_factor, scale)
            + _dynarray(data))


def read_ciphertext_members(m: bytes, off: int = 0):
    pid = struct.unpack_from("<4Q", m, off)
    is_ntt, size, N, limbs, cf, scale = struct.unpack_from("<BQQQQd", m, off + 32)
    data, off = _read_dynarray(m, off + 32 + 41)
    return {"parms_id": pid, "is_ntt": bool(is_ntt), "size": size, "N": N, "limbs": limbs, "correction_factor": cf, "scale": scale,
            "data": data.reshape(size, limbs, N)}, off


# ---- Plaintext / SecretKey -------------------------------------------------------------------------------------------------------
def plaintext_members(pid, data: np.ndarray, scale: float = 1.0) -> bytes:
    return struct.pack("<4Q", *pid) + struct.pack("<Qd", data.size, scale) + _dynarray(data)


def read_plaintext_members(m: bytes):
    pid = struct.unpack_from("<4Q", m, 0)
    n, scale = struct.unpack_from("<Qd", m, 32)
    data, _ = _read_dynarray(m, 48)
    assert data.size == n
    return {"parms_id": pid, "coeff_count": n, "scale": scale, "data": data}


# ---- KSwitchKeys (RelinKeys / GaloisKeys) ----------------------------------------------------------------------------------------
def kswitch_members(pid, dim1: int, present: dict) -> bytes:
    """present: index -> key array [digits][2][K][N]"""
    out = [struct.pack("<4Q", *pid), struct.pack("<Q", dim1)]
    for index in range(dim1):
        key = present.get(index)
        if key is None:
            out.append(struct.pack("<Q", 0))
            continue
        out.append(struct.pack("<Q", key.shape[0]))
        for j in range(key.shape[0]):
            out.append(wrap(ciphertext_members(pid, key[j])))
    return b"".join(out)


def read_kswitch_members(m: bytes):
    pid = struct.unpack_from("<4Q", m, 0)
    (dim1,) = struct.unpack_from("<Q", m, 32)
    off, present = 40, {}
    for index in range(dim1):
        (dim2,) = struct.unpack_from("<Q", m, off)
        off += 8
        digits = []
        for _ in range(dim2):
            body, off = unwrap(m, off)
            ct, _ = read_ciphertext_members(body)
            assert ct["parms_id"] == pid and ct["size"] == 2
            digits.append(ct["data"])
        if digits:
            present[index] = np.stack(digits)
    return {"parms_id": pid, "dim1": dim1, "present": present}


# ---- the five files of SEAL_HEVM::create_context ------------------------------------------------------------------------------------
def write_key_dir(path, N: int, primes, pk, sk, relin, galois: dict, compr: int = COMPR_NONE):
    """pk [2][K][N], sk [K][N], relin [K-1][2][K][N], galois: elt -> [K-1][2][K][N]  (all NTT form, SEAL limb order)"""
    path = Path(path)
    kid = parms_id(N, primes)
    (path / "parm.seal").write_bytes(wrap(params_members(N, primes), compr))
    (path / "pub.seal").write_bytes(wrap(ciphertext_members(kid, np.asarray(pk)), compr))
    (path / "sec.seal").write_bytes(wrap(plaintext_members(kid, np.asarray(sk)), compr))
    (path / "relin.seal").write_bytes(wrap(kswitch_members(kid, 1, {0: np.asarray(relin)}), compr))
    (path / "gal.seal").write_bytes(wrap(kswitch_members(kid, N, {(int(e) - 1) >> 1: np.asarray(k) for e, k in galois.items()}), compr))


def read_key_dir(path):
    path = Path(path)
    out = {"params": read_params_members(unwrap((path / "parm.seal").read_bytes())[0])}
    if (path / "pub.seal").exists():
        out["pk"] = read_ciphertext_members(unwrap((path / "pub.seal").read_bytes())[0])[0]
    if (path / "sec.seal").exists():
        out["sk"] = read_plaintext_members(unwrap((path / "sec.seal").read_bytes())[0])
    if (path / "relin.seal").exists():
        out["relin"] = read_kswitch_members(unwrap((path / "relin.seal").read_bytes())[0])
    if (path / "gal.seal").exists():
        g = read_kswitch_members(unwrap((path / "gal.seal").read_bytes())[0])
        g["by_elt"] = {2 * i + 1: k for i, k in g["present"].items()}
        out["gal"] = g
    return out
