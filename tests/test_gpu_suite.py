"""GPU: the reference's benchmark suite (examples/benchmarks/*.py traced into HEVM programs, tests/golden/suite/*) through
the HEVM boundary at the reference's parameters, with the inputs the reference's examples/tests/*.py scripts feed:
decrypted results against the cleartext evaluation, and -- for the programs without opcode 10 -- final ciphertext limbs
bit-identical to the oracle VM on the same key / plaintext / input limbs."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.oracle import Oracle

from gpu_helpers import _get_ct, _import_keys, _mirror_vm

SUITE = Path(__file__).resolve().parent / "golden" / "suite"
NAMES = ["SobelFilter", "HarrisCornerDetection", "LinearRegression", "PolynomialRegression", "Multivariate", "MLP"]


@pytest.fixture(scope="module")
def vm15o():
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D + 2, logN=15, num_primes=14)
    o = Oracle(15, 14)
    _import_keys(o, hevm, ll)
    return hevm, o, ll


@pytest.mark.parametrize("name", NAMES)
def test_suite_program(vm15o, name, tmp_path):
    from dacapo_amd import hevm_asm as ha

    hevm, o, ll = vm15o
    fx = ha.read_fixture(SUITE / name)
    hevm.load_mem(fx["cst"], fx["hevm"])
    assert hevm.arglen == len(fx["inputs"]) and hevm.reslen == fx["expected"].shape[0]
    for i, x in enumerate(fx["inputs"]):
        hevm.setInput(i, x)
    deterministic = fx["meta"]["bootstraps"] == 0
    if deterministic:
        ovm = _mirror_vm(hevm, ll, o, fx["cst"], fx["hevm"], tmp_path)
        for i in range(hevm.arglen):
            ovm.ciphers[i] = _get_ct(hevm, ll, i)
    hevm.run()
    res = hevm.getOutput()
    want = fx["expected"]
    scale = max(1.0, float(np.abs(want).max()))
    rms = float(np.sqrt(np.mean((res - want) ** 2)))
    print(f"{name}: rms vs cleartext evaluation {rms:.3e} (magnitude {scale:.3g})")
    assert rms < 2e-4 * scale
    if deterministic:
        ovm.run()
        for r in ovm.prog.res_dst:
            got, exp = _get_ct(hevm, ll, r), ovm.ciphers[r]
            assert got.ell == exp.ell and got.scale == exp.scale and (got.data == exp.data).all()
