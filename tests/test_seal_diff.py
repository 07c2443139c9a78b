"""Differential test against Microsoft SEAL itself.  It runs on COMMITTED SEAL output when tests/golden/seal_vectors/ exists (written once, on
any machine that has SEAL 4.x, by `SEAL_ROOT=... bash tools/fixtures/make_seal_vectors.sh`: from then on no SEAL is needed, here or on the GPU
box); otherwise on a scenario generated on the spot when a SEAL 4.x install is found (SEAL_ROOT); otherwise it is SKIPPED (no SEAL exists in
the build image or on the GPU box: SURVEY.md 8c).  It closes the loop that "parity unpinned" leaves open: tools/fixtures/seal_diff_gen.cpp runs SEAL on a fixed scenario and saves keys, inputs and every result in
SEAL's serialization; the oracle (CPU test) and the MI355X runtime (GPU test, through initFullVM / hevm_load_ctxt) replay the
same evaluator calls on the same keys and inputs and must produce the same limbs, bit for bit."""
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _find_seal():
    roots = [os.environ.get("SEAL_ROOT"), "/usr/local", "/usr", "/opt/seal", str(Path.home() / "mylibs"), str(Path.home() / ".local")]
    for r in [x for x in roots if x]:
        for inc in sorted(Path(r).glob("include/SEAL-4.*")):
            if (inc / "seal" / "seal.h").exists():
                for lib in list(Path(r).glob("lib*/libseal-4.*.a")) + list(Path(r).glob("lib*/libseal*.so")):
                    return inc, lib
    return None


VECTORS = ROOT / "tests" / "golden" / "seal_vectors"


@pytest.fixture(scope="module")
def scenario(tmp_path_factory):
    if (VECTORS / "MANIFEST.json").exists() and (VECTORS / "parm.seal").exists():  # SEAL's own output, committed: preferred, needs no SEAL
        import hashlib
        import json

        man = json.loads((VECTORS / "MANIFEST.json").read_text())
        for name, rec in man["files"].items():
            assert hashlib.sha256((VECTORS / name).read_bytes()).hexdigest() == rec["sha256"], f"{name} differs from what SEAL wrote"
        return VECTORS
    found = _find_seal()
    if found is None or shutil.which("g++") is None:
        pytest.skip("Microsoft SEAL 4.x not installed (set SEAL_ROOT): the SEAL-side pin of the arithmetic cannot run here")
    inc, lib = found
    d = tmp_path_factory.mktemp("seal_diff")
    exe = d / "seal_diff_gen"
    base = ["g++", "-std=c++17", "-O2", str(ROOT / "tools" / "fixtures" / "seal_diff_gen.cpp"), f"-I{inc}", "-o", str(exe)]
    for extra in ([str(lib)], [str(lib), "-lzstd", "-lz"], [f"-L{lib.parent}", "-lseal-4.0", "-lzstd", "-lz"]):
        if subprocess.run(base + extra + ["-lpthread"], capture_output=True).returncode == 0:
            break
    else:
        pytest.skip("SEAL found but the generator does not link against it")
    out = d / "out"
    out.mkdir()
    subprocess.run([str(exe), str(out), "13", "5"], check=True, env=dict(os.environ, LD_LIBRARY_PATH=str(lib.parent)))
    return out


CASES = ["rotate_1", "rotate_37", "rotate_-100", "negate", "add", "modswitch", "mul", "rescale"]


def _read_ct(path):
    from oracle import seal_format as sf

    raw = path.read_bytes()
    if raw[5] == sf.COMPR_ZSTD:
        pytest.skip("scenario saved with Zstandard and no Python zstd reader is available; rebuild SEAL with -DSEAL_USE_ZSTD=OFF "
                    "or read through the product (GPU test)")
    return sf.read_ciphertext_members(sf.unwrap(raw)[0])[0]


def test_oracle_reproduces_seal_limbs(scenario):
    from oracle import seal_format as sf
    from oracle.oracle import Ciphertext, Oracle

    keys = sf.read_key_dir(scenario)
    N, primes = keys["params"]["N"], keys["params"]["primes"]
    o = Oracle(N.bit_length() - 1, len(primes))
    assert o.primes == primes                                   # CoeffModulus::Create order
    K = len(primes)
    o.sk, o.pk = keys["sk"]["data"].reshape(K, N), keys["pk"]["data"]
    o.relin, o.galois = keys["relin"]["present"][0], keys["gal"]["by_elt"]
    assert sorted(o.galois) == sorted(set(o.default_galois_elts()))
    A, B = (_read_ct(scenario / f"{n}.ct") for n in "ab")
    assert A["parms_id"] == sf.parms_id(N, primes[:K - 1])      # first data level
    a, b = Ciphertext(A["data"], A["scale"]), Ciphertext(B["data"], B["scale"])
    m = o.mul_relin(a, b)
    got = {"rotate_1": o.rotate(a, 1), "rotate_37": o.rotate(a, 37), "rotate_-100": o.rotate(a, -100), "negate": o.negate(a),
           "add": o.add(a, b), "modswitch": o.modswitch(a, 1), "mul": m, "rescale": o.rescale(m)}
    for name in CASES:
        want = _read_ct(scenario / f"{name}.ct")
        assert want["limbs"] == got[name].ell and want["scale"] == got[name].scale, name
        assert (want["data"] == got[name].data).all(), name      # bit-identical to SEAL
    # encoder: the FFT is floating point; a coefficient may differ by one unit
    pa = sf.read_plaintext_members(sf.unwrap((scenario / "a.pt").read_bytes())[0])
    vec = np.fromfile(scenario / "a.f64")
    mine = o.encode(vec, 2.0**40, K - 1)
    d = o.ntt_inv(mine.data, list(range(K - 1))).astype(np.int64) - o.ntt_inv(pa["data"].reshape(K - 1, N), list(range(K - 1))).astype(np.int64)
    assert np.abs(d).max() <= 1
    # and SEAL's decryption of the product decodes to a*b
    dec = np.fromfile(scenario / "mul.decoded.f64")
    assert np.abs(dec[: N // 2] - vec * np.fromfile(scenario / "b.f64")).max() < 1e-4


@pytest.mark.gpu
def test_gpu_runtime_reproduces_seal_limbs(scenario, tmp_path):
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import runner
    from oracle import seal_format as sf

    hevm = runner.HEVM(path=str(scenario))                      # SEAL-written key directory through initFullVM
    K = hevm.K
    R, NEG, RS, MS, ADD, MUL = ha.OP_ROTATE, ha.OP_NEGATE, ha.OP_RESCALE, ha.OP_MODSWITCH, ha.OP_ADDCC, ha.OP_MULCC
    ops = [(R, 2, 0, 1), (R, 3, 0, 37), (R, 4, 0, (-100) & 0xFFFF), (NEG, 5, 0, 0), (ADD, 6, 0, 1), (MS, 7, 0, 1), (MUL, 8, 0, 1), (RS, 9, 8, 0)]
    lv = K - 1
    hv = ha.pack_hevm([40, 40], [lv, lv], [40] * 8, [lv] * 8, list(range(2, 10)), 10, 0, lv, np.array(ops, dtype=np.uint16))
    hevm.load_mem(ha.pack_cst([]), hv)
    hevm.loadCtxt(0, scenario / "a.ct")
    hevm.loadCtxt(1, scenario / "b.ct")
    hevm.run()
    for reg, name in zip(range(2, 10), CASES):
        hevm.saveCtxt(reg, tmp_path / "got.ct")
        got = sf.read_ciphertext_members(sf.unwrap((tmp_path / "got.ct").read_bytes())[0])[0]
        hevm.loadCtxt(15, scenario / f"{name}.ct")              # the product's reader also handles zstd
        hevm.saveCtxt(15, tmp_path / "want.ct")
        want = sf.read_ciphertext_members(sf.unwrap((tmp_path / "want.ct").read_bytes())[0])[0]
        assert got["limbs"] == want["limbs"] and got["scale"] == want["scale"] and got["parms_id"] == want["parms_id"], name
        assert (got["data"] == want["data"]).all(), name        # bit-identical to SEAL
