"""GPU (one box): what the 8-GPU configuration rests on (one VM replica per GPU, keys replicated) --
  * two VMs created from the same seed hold the same key set (equal device-side digests, equal buffer lists), a third from another
    seed does not;
  * given the same input they produce IDENTICAL result limbs for a program with opcode 10 in it (the encryption randomness is
    expanded from the seed too), i.e. any replica can serve any of the client's ciphertext streams;
  * a replica whose keys were overwritten with another's buffers (what bench.py --broadcast-keys does over RCCL) ends up with that
    replica's digest and results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _prog(slots):
    from dacapo_amd import hevm_asm as ha

    rng = np.random.default_rng(4)
    b = ha.Builder(slots=slots, init_level=3, policy="lazy", boot_level=3, shadow=True)
    x = b.input(rng.uniform(-1, 1, slots))
    y = x
    for _ in range(4):                                   # runs out of primes: opcode 10 appears
        y = b.add_plain(b.mul_plain(b.mul(y, y), [0.5]), [0.1])
    b.output(b.finish(b.add(y, b.rotate(x, 3))))
    cst, hv, info = b.assemble()
    assert info["op_mix"]["bootstrap"] >= 1
    return b, cst, hv


def _result_limbs(hevm, ll):
    c = hevm.getCtxt(hevm.getResIdx(0))
    return ll.read_device(c.data, (2, c.poly_stride // hevm.N, hevm.N))[:, : c.level]


def test_same_seed_replicas_agree_limb_for_limb_and_keys_can_be_shipped():
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    logN, K = 12, 5
    b, cst, hv = _prog(1 << (logN - 1))
    vms = [runner.HEVM(seed=s, logN=logN, num_primes=K) for s in (31, 31, 32)]
    d = [v.keyDigest() for v in vms]
    assert d[0] == d[1] and d[0] != d[2]
    sizes = [[w for _, w in v.keyBuffers()] for v in vms]
    assert sizes[0] == sizes[1] == sizes[2] and len(sizes[0]) >= 3 + 20           # sk, pk, relin + the default Galois keys
    outs = []
    for v in vms:
        v.load_mem(cst, hv)
        v.setInput(0, b.args[0].plain)
        v.run()
        outs.append(_result_limbs(v, ll))
        assert np.abs(v.getOutput()[0] - b.expected()[0]).max() < 1e-4
    assert (outs[0] == outs[1]).all()                                             # same seed -> same keys, same randomness, same limbs
    assert not (outs[0] == outs[2]).all()
    # ship replica 0's keys into replica 2 (device-to-device here; bench.py --broadcast-keys does it with dist.broadcast)
    L = ll.lib()
    for (dst, w), (src, w2) in zip(vms[2].keyBuffers(), vms[0].keyBuffers()):
        assert w == w2
        L.dc_memcpy_d2d(dst, src, 8 * w, None)
    L.dc_device_sync()
    vms[2].keysReplaced()
    assert vms[2].keyDigest() == d[0]
    vms[2].load_mem(cst, hv)
    vms[2].setInput(0, b.args[0].plain)
    vms[2].run()
    assert np.abs(vms[2].getOutput()[0] - b.expected()[0]).max() < 1e-4           # decrypts under the shipped secret key
    for v in vms:
        v.close()
