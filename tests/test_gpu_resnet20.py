"""GPU: the reference's ResNet-20 (examples/benchmarks/ResNet.py traced with its resnet20.silu.model weights; committed
as data under tests/golden/resnet20.*, see tools/fixtures/trace_reference_model.py) through the HEVM boundary at the reference's
parameters (N = 2^15, 14 x 60-bit primes):
  * the first layer (everything before the first opcode 10, whose fresh randomness the oracle cannot reproduce) is
    bit-identical to the oracle VM on the same key/plaintext/input limbs;
  * the whole encrypted inference decrypts to the torch model's logits (what examples/tests/ResNet.py checks)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from pathlib import Path

from oracle.oracle import Oracle

from gpu_helpers import _get_ct, _import_keys, _mirror_vm

GOLDEN = Path(__file__).resolve().parent / "golden" / "resnet20"


@pytest.fixture(scope="module")
def fixture20():
    from dacapo_amd import hevm_asm as ha

    return ha.read_fixture(GOLDEN)


@pytest.fixture(scope="module")
def vm15():
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    return runner.HEVM(seed=0x4845564D, logN=15, num_primes=14), ll


def test_resnet20_first_layer_bit_exact(vm15, fixture20, tmp_path):
    from dacapo_amd import hevm_asm as ha

    hevm, ll = vm15
    ops = ha.unpack_hevm(fixture20["hevm"])["ops"]
    first_boot = int(np.nonzero(ops[:, 0] == ha.OP_BOOTSTRAP)[0][0])
    hv, lvl, scale_bits = ha.truncate_hevm(fixture20["hevm"], first_boot)
    assert (ops[:first_boot, 0] == ha.OP_ROTATE).sum() >= 20  # the stem convolution's rotations are in the prefix
    hevm.load_mem(fixture20["cst"], hv)
    o = Oracle(15, 14)
    _import_keys(o, hevm, ll)
    ovm = _mirror_vm(hevm, ll, o, fixture20["cst"], hv, tmp_path)
    hevm.setInput(0, fixture20["packed"])
    ovm.ciphers[0] = _get_ct(hevm, ll, 0)
    hevm.run()
    ovm.run()
    reg = ovm.prog.res_dst[0]
    got, want = _get_ct(hevm, ll, reg), ovm.ciphers[reg]
    assert got.ell == want.ell == lvl and got.scale == want.scale
    assert (got.data == want.data).all()


def test_resnet20_explicitly_built_graph(fixture20):
    """option plan_graph = 2: the plan's graph built from its own dependencies (HEVM::capture_plan_dag) -- the first layer limb for limb like
    the default, the whole inference to the same logits, twice (a replayed graph re-encrypts freshly)"""
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D + 2, logN=15, num_primes=14, vm_options={"plan_graph": 2})
    hevm.load_mem(fixture20["cst"], fixture20["hevm"])
    hevm.setInput(0, fixture20["packed"])
    for _ in range(2):
        hevm.run()
        out = hevm.getOutput()[0]
        assert float(np.sqrt(np.mean((out - fixture20["expected"]) ** 2))) < 1e-3
        assert float(np.sqrt(np.mean((out[:10] * 32 - fixture20["torch_result"]) ** 2))) < 2e-3
    hevm.close()


def test_resnet20_single_stream_no_graph(fixture20):
    """the plan issued launch by launch on one stream (the default replays it as one HIP graph with an auxiliary stream)"""
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D + 1, logN=15, num_primes=14, vm_options={"plan_lanes": 1, "plan_graph": 0})
    hevm.load_mem(fixture20["cst"], fixture20["hevm"])
    hevm.setInput(0, fixture20["packed"])
    for _ in range(2):
        hevm.run()
        out = hevm.getOutput()[0]
        assert float(np.sqrt(np.mean((out - fixture20["expected"]) ** 2))) < 1e-3


def test_resnet20_rescale_folded_into_opcode10(fixture20):
    """opt-in: a rescale that only feeds an opcode 10 is done inside its re-encoder; same logits"""
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D + 4, logN=15, num_primes=14, vm_options={"fold_rescale_boot": 1})
    hevm.load_mem(fixture20["cst"], fixture20["hevm"])
    hevm.setInput(0, fixture20["packed"])
    hevm.run()
    out = hevm.getOutput()[0]
    assert float(np.sqrt(np.mean((out - fixture20["expected"]) ** 2))) < 1e-3
    assert hevm.stats()["ntts"] < 55280  # the folded rescales' NTTs are neither executed nor counted


def test_resnet20_two_ciphertext_streams(fixture20):
    """throughput mode: two independent images through one plan (every batched step carries both streams' items, every
    opcode 10 its own zero-encryption)"""
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D + 3, logN=15, num_primes=14)
    hevm.set_streams(2)
    hevm.load_mem(fixture20["cst"], fixture20["hevm"])
    flipped = fixture20["packed"][::-1].copy()  # a different (meaningless) image for stream 1: same program, other data
    from dacapo_amd import hevm_asm as ha

    want1 = ha.plain_eval(fixture20["hevm"], fixture20["cst"], [flipped * 0.5])[0]
    hevm.select_stream(0)
    hevm.setInput(0, fixture20["packed"])
    hevm.select_stream(1)
    hevm.setInput(0, flipped * 0.5)
    hevm.run()
    hevm.select_stream(0)
    out0 = hevm.getOutput()[0]
    hevm.select_stream(1)
    out1 = hevm.getOutput()[0]
    assert float(np.sqrt(np.mean((out0 - fixture20["expected"]) ** 2))) < 1e-3
    assert float(np.sqrt(np.mean((out1 - want1) ** 2))) < 1e-3 * max(1.0, float(np.abs(want1).max()))


def test_resnet20_encrypted_inference_matches_torch(vm15, fixture20):
    hevm, ll = vm15
    hevm.load_mem(fixture20["cst"], fixture20["hevm"])
    hevm.setInput(0, fixture20["packed"])
    hevm.run()
    out = hevm.getOutput()[0]
    st = hevm.stats()
    mix = fixture20["meta"]["info"]["op_mix"]
    assert st["op_counts"][1] == mix["rotate"] and st["op_counts"][8] == mix["mulcc"] and st["op_counts"][10] == mix["bootstrap"]
    logits = out[:10] * 32  # examples/tests/ResNet.py:76-81
    want = fixture20["torch_result"]
    assert int(np.argmax(logits)) == int(np.argmax(want))
    rms_torch = float(np.sqrt(np.mean((logits - want) ** 2)))
    rms_plain = float(np.sqrt(np.mean((out - fixture20["expected"]) ** 2)))
    print(f"ResNet-20 on MI355X: rms vs torch {rms_torch:.3e} (reference README: 9.5e-4), vs plaintext evaluation {rms_plain:.3e}")
    # the polynomial SiLU alone is 5.4e-4 away from torch on this image (golden/resnet20.json plain_vs_torch_rms)
    assert rms_torch < 3e-3
    assert rms_plain < 1e-3
    # a second run() re-executes the whole program on the same inputs (fresh bootstrap randomness): same answer
    hevm.run()
    again = hevm.getOutput()[0]
    assert np.abs(again - out).max() < 1e-3


@pytest.mark.parametrize("tag", ["b6", "b13"])
def test_resnet20_other_lowerings_decrypt_to_the_same_logits(fixture20, tag):
    """the same trace with opcode 10 re-encrypting to 6 / 13 primes (key switches at up to 13 primes, the reference's top level):
    the encrypted inference still decrypts to the torch model's logits"""
    import gzip
    from pathlib import Path

    from dacapo_amd import runner

    hv = gzip.open(Path(__file__).resolve().parent / "golden" / f"resnet20.{tag}.hevm.gz").read()
    hevm = runner.HEVM(seed=0x4845564D + 9, logN=15, num_primes=14)
    hevm.load_mem(fixture20["cst"], hv)
    hevm.setInput(0, fixture20["packed"])
    hevm.run()
    out = hevm.getOutput()[0]
    assert float(np.sqrt(np.mean((out - fixture20["expected"]) ** 2))) < 1e-3
    assert float(np.sqrt(np.mean((out[:10] * 32 - fixture20["torch_result"]) ** 2))) < 5e-3


def test_resnet20_online_encode_shrinks_the_plaintext_footprint(fixture20):
    """option online_encode = 1: the 5 894 plaintext registers are encoded at use from the resident constants (0.5 GB of doubles)
    into a recycled window instead of living pre-encoded in HBM (6 GB); same logits"""
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D + 6, logN=15, num_primes=14, vm_options={"online_encode": 1})
    hevm.load_mem(fixture20["cst"], fixture20["hevm"])
    hevm.setInput(0, fixture20["packed"])
    hevm.run()
    out = hevm.getOutput()[0]
    assert float(np.sqrt(np.mean((out - fixture20["expected"]) ** 2))) < 1e-3
    assert hevm.plaintextBytes() < 2.0e9      # constants 0.49 GB + window + scratch = 1.7 GB, against 4 GB pre-encoded
