"""GPU: SEAL 4.0 serialization at the drop-in boundary (SEAL_HEVM.cpp:55-88 create_context, :91-180 loadSEAL / loadClient /
loadServer) and the client / server split those three init symbols exist for (:405-419, :431-437).

  * create_context writes SEAL-format files that the independent Python reader (oracle/seal_format.py) parses; their limbs are
    the VM's keys and satisfy the RLWE key equations under the oracle's arithmetic;
  * a key directory written by the ORACLE side (own keygen, own writer, zlib-compressed) loads through initFullVM and runs a
    program bit-identically to the oracle VM;
  * ciphertexts travel as seal::Ciphertext bytes between a client VM (pk + sk) and a server VM (relin + Galois keys only);
  * the samplers' ChaCha20 block function on the device equals the host's (tests/test_seal_format.py pins that to RFC 8439)."""
import ctypes as C
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import seal_format as sf
from oracle.oracle import Ciphertext, Oracle, OracleVM, Plaintext


def _dev(ll, ptr, shape):
    return ll.read_device(ptr, shape)


@pytest.fixture(scope="module")
def keydir(tmp_path_factory):
    """create_context at N = 2^12, 4 primes (the reference hard-codes 2^15 / 14; options logn / primes shrink it for tests)"""
    from dacapo_amd import runner

    d = tmp_path_factory.mktemp("seal_keys")
    with runner.options(logn=12, primes=4):
        runner.lw.create_context(str(d).encode())
    return d


def test_chacha20_on_the_device_equals_the_host(keydir):
    from dacapo_amd import runner

    lw = runner.reinit_lw()
    lw.hevm_chacha20_block.argtypes = [C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    lw.hevm_chacha20_blocks_device.argtypes = [C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    key = (C.c_uint32 * 8)(*range(0x01020304, 0x01020304 + 8))
    blocks = 200
    out = np.zeros((blocks, 16), dtype=np.uint32)
    lw.hevm_chacha20_blocks_device(key, (7 << 20) | 5, 0x1234_0000_0106, blocks, out.ctypes.data)
    for b in (0, 1, 63, 64, 199):
        want = (C.c_uint32 * 16)()
        lw.hevm_chacha20_block(key, ((7 << 20) | 5) + b, 0x1234_0000_0106, want)
        assert out[b].tolist() == list(want)


def test_created_key_directory_is_seal_format(keydir):
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    got = sf.read_key_dir(keydir)
    o = Oracle(12, 4)
    N, K = 1 << 12, 4
    assert got["params"] == {"scheme": 2, "N": N, "primes": o.primes, "plain_modulus": 0}
    kid = sf.parms_id(N, o.primes)
    hevm = runner.HEVM(path=str(keydir))  # initFullVM on the same directory
    lw = runner.lw
    assert got["pk"]["parms_id"] == kid and got["pk"]["size"] == 2 and got["pk"]["is_ntt"] and got["pk"]["scale"] == 1.0
    assert (got["pk"]["data"] == _dev(ll, lw.hevm_public_key(hevm.vm), (2, K, N))).all()
    assert got["sk"]["parms_id"] == kid and got["sk"]["coeff_count"] == K * N
    assert (got["sk"]["data"].reshape(K, N) == _dev(ll, lw.hevm_secret_key(hevm.vm), (K, N))).all()
    assert got["relin"]["dim1"] == 1 and list(got["relin"]["present"]) == [0]
    assert (got["relin"]["present"][0] == _dev(ll, lw.hevm_relin_key(hevm.vm), (K - 1, 2, K, N))).all()
    # GaloisKeys: N slots, KeyGenerator::create_galois_keys' default set (3^(+-2^k), conjugation) = 2 (logN - 1) + 1 keys
    assert got["gal"]["dim1"] == N and sorted(got["gal"]["by_elt"]) == sorted(set(o.default_galois_elts()))  # 3^(2^(logN-2)) is its own inverse: listed twice, generated once
    for elt, key in got["gal"]["by_elt"].items():
        assert (key == _dev(ll, lw.hevm_galois_key(hevm.vm, elt), (K - 1, 2, K, N))).all()
    # the file contents are keys: pk decrypts to small noise, the secret is ternary, a Galois key satisfies its equation
    o.sk, o.pk = got["sk"]["data"].reshape(K, N), got["pk"]["data"]
    s = o.ntt_inv(o.sk, list(range(K)))
    assert all(set(np.unique(s[i]).tolist()) <= {0, 1, q - 1} for i, q in enumerate(o.primes))
    assert len(np.unique(s[0])) == 3
    e = o.ntt_inv(o.poly_add(o.pk[0], o.poly_mul(o.pk[1], o.sk)), list(range(K)))[0].astype(np.int64)
    e = np.where(e > o.primes[0] // 2, e - np.int64(o.primes[0]), e)
    assert np.abs(e).max() <= 21 and e.std() > 2.0


def test_two_contexts_never_share_randomness(keydir, tmp_path):
    """create_context draws from getrandom(): two directories made by the same binary have unrelated keys"""
    from dacapo_amd import runner

    with runner.options(logn=12, primes=4):
        runner.lw.create_context(str(tmp_path).encode())
    a, b = sf.read_key_dir(keydir), sf.read_key_dir(tmp_path)
    assert (a["sk"]["data"] != b["sk"]["data"]).mean() > 0.4 and (a["pk"]["data"][1] != b["pk"]["data"][1]).mean() > 0.99


def _program(slots, init_level=3):
    from dacapo_amd import hevm_asm as ha

    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, slots)
    b = ha.Builder(slots=slots, init_level=init_level)
    v = b.input(x)
    y = b.add(b.mul(v, b.rotate(v, 5)), b.mul_plain(b.rotate(v, -64), rng.uniform(-1, 1, slots)))
    y = b.add_plain(b.mul(y, v), [0.125])
    b.output(y)
    return b, x


def test_key_directory_written_by_the_oracle_side_drives_the_vm(tmp_path):
    """independent keygen + independent writer -> initFullVM -> the program's result limbs equal the oracle VM's"""
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner
    from gpu_helpers import _get_ct, _mirror_vm

    o = Oracle(12, 4)
    o.keygen(seed=99)
    keys = tmp_path / "keys"
    keys.mkdir()
    sf.write_key_dir(keys, o.N, o.primes, o.pk, o.sk, o.relin, o.galois, compr=sf.COMPR_ZLIB)
    hevm = runner.HEVM(path=str(keys))
    assert hevm.logN == 12 and hevm.K == 4
    b, x = _program(o.slots)
    cst, hv, _ = b.assemble()
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    hevm.setInput(0, x)
    ovm.ciphers[0] = _get_ct(hevm, ll, 0)
    assert np.abs(o.decode(o.decrypt(ovm.ciphers[0])) - x).max() < 1e-6  # encrypted under the oracle's pk, decrypts under its sk
    hevm.run()
    ovm.run()
    r = ovm.prog.res_dst[0]
    got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
    assert got.ell == want.ell and got.scale == want.scale and (got.data == want.data).all()
    assert np.sqrt(np.mean((hevm.getOutput()[0] - b.expected()[0]) ** 2)) < 1e-4


def test_ciphertext_files_round_trip_and_carry_the_level_id(keydir, tmp_path):
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner
    from gpu_helpers import _get_ct

    hevm = runner.HEVM(path=str(keydir))
    b, x = _program(hevm.slots)
    cst, hv, _ = b.assemble()
    hevm.load_mem(cst, hv)
    hevm.setInput(0, x)
    o = Oracle(12, 4)
    for compr in (0, 1):                                                 # option seal_compr: none, zlib
        with runner.options(seal_compr=compr):
            hevm.saveCtxt(0, tmp_path / "in.ct")
        ct, _ = sf.read_ciphertext_members(sf.unwrap((tmp_path / "in.ct").read_bytes())[0])
        dev = _get_ct(hevm, ll, 0)
        assert ct["limbs"] == 3 and ct["parms_id"] == sf.parms_id(o.N, o.primes[:3]) and ct["scale"] == 2.0**40
        assert (ct["data"] == dev.data).all()
    # a file written by the Python side loads into another register and decrypts to the same slots
    (tmp_path / "py.ct").write_bytes(sf.wrap(sf.ciphertext_members(ct["parms_id"], ct["data"], scale=ct["scale"]), sf.COMPR_ZLIB))
    hevm.loadCtxt(5, tmp_path / "py.ct")
    again = _get_ct(hevm, ll, 5)
    assert again.ell == 3 and again.scale == 2.0**40 and (again.data == dev.data).all()


def test_client_and_server_vms_exchange_seal_ciphertexts(keydir, tmp_path):
    """initClientVM: pk + sk, encrypts and decrypts, never sees the program body; initServerVM: relin + Galois keys only, runs
    it (SEAL_HEVM.cpp:410-419, :431-437).  What crosses between them is seal::Ciphertext bytes."""
    from dacapo_amd import runner

    b, x = _program(1 << 11)
    b.write(tmp_path / "_hecate_p.cst", tmp_path / "p.40._hecate_p.hevm")
    client = runner.HEVM(path=str(keydir), option="client")
    client.load(str(tmp_path / "_hecate_p.cst"), str(tmp_path / "p.40._hecate_p.hevm"))
    assert client.arglen == 1 and client.reslen == 1
    client.setInput(0, x)
    client.saveCtxt(0, tmp_path / "arg0.ct")

    server = runner.HEVM(path=str(keydir), option="server")
    lw = runner.lw
    assert not lw.hevm_secret_key(server.vm) and not lw.hevm_public_key(server.vm) and lw.hevm_relin_key(server.vm)
    server.load(str(tmp_path / "_hecate_p.cst"), str(tmp_path / "p.40._hecate_p.hevm"))
    server.loadCtxt(0, tmp_path / "arg0.ct")
    server.run()
    server.saveCtxt(server.getResIdx(0), tmp_path / "res0.ct")

    client.loadCtxt(client.getResIdx(0), tmp_path / "res0.ct")  # loadClient placed result i in register arg_len + i
    res = client.getOutput()[0]
    assert np.sqrt(np.mean((res - b.expected()[0]) ** 2)) < 1e-4


def test_server_vm_aborts_cleanly_on_opcode_10(keydir, tmp_path):
    """the SEAL VM's bootstrap stand-in decrypts: a server VM (no secret key) must stop with a message, not compute garbage"""
    from dacapo_amd import hevm_asm as ha

    b = ha.Builder(slots=1 << 11, init_level=3)
    v = b.input(np.linspace(-1, 1, 1 << 11))
    b.output(b.bootstrap(b.mul(v, v), 3))
    b.write(tmp_path / "_hecate_b.cst", tmp_path / "b.40._hecate_b.hevm")
    code = f"""
import sys
sys.path.insert(0, {str(__import__('pathlib').Path(__file__).resolve().parent.parent)!r})
from dacapo_amd import runner
s = runner.HEVM(path={str(keydir)!r}, option="server")
s.load({str(tmp_path / '_hecate_b.cst')!r}, {str(tmp_path / 'b.40._hecate_b.hevm')!r})
c = runner.HEVM(path={str(keydir)!r}, option="client")
c.load({str(tmp_path / '_hecate_b.cst')!r}, {str(tmp_path / 'b.40._hecate_b.hevm')!r})
import numpy as np
c.setInput(0, np.linspace(-1, 1, 1 << 11))
c.saveCtxt(0, {str(tmp_path / 'a.ct')!r})
s.loadCtxt(0, {str(tmp_path / 'a.ct')!r})
print("before run", flush=True)
s.run()
print("after run", flush=True)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "before run" in r.stdout and "after run" not in r.stdout and r.returncode != 0
    assert "bootstrap" in r.stderr and ("secret" in r.stderr or "public key" in r.stderr)
