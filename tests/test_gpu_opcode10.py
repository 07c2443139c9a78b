"""GPU: the deterministic half of opcode 10.  The SEAL VM's "bootstrap" is decrypt -> decode -> encode(scale 2^floor(log2 scale),
target level) -> encrypt (/root/reference/lib/Runtime/SEAL_HEVM.cpp:328-333).  The device path never runs an FFT: it evaluates
decode o encode as the projection m' = round((m + m(X^-1))/2 * new_scale/old_scale) on the decrypted polynomial.
With the test hook that makes encryptions of zero (0, 0), the result ciphertext is (re-encoded plaintext, 0), so that plaintext
can be compared coefficient by coefficient with the oracle's decrypt -> decode -> encode (double-precision FFTs, like SEAL's).

Bound, stated: at the working point of the traced ResNet-20 (source at 1 prime and scale ~2^40, |coefficients| < 2^53, every
integer exactly representable) the two paths differ by at most ONE unit in a coefficient -- the oracle's FFT round-off
(~1e-3 units) can flip a rounding at a half-integer; at scale 2^80 a double holds 53 of the ~80 bits, so both paths are only
accurate to 2^-50 of the scale and the bound is relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gpu_helpers import _get_ct, _import_keys  # noqa: E402
from oracle.oracle import Oracle, Plaintext  # noqa: E402


def _centered_diff(o, a, b, ell):
    ca, cb = o.ntt_inv(a, list(range(ell))), o.ntt_inv(b, list(range(ell)))
    q = np.uint64(o.primes[0])
    d = (ca[0] + (q - cb[0])) % q                      # difference mod q_0, centred
    d = d.astype(np.int64)
    return np.where(d > int(q) // 2, d - np.int64(int(q)), d)


@pytest.mark.parametrize("plan", [1, 0])
def test_reencoded_plaintext_matches_decode_then_encode(plan):
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D, logN=13, num_primes=7, vm_options={"plan": plan})
    o = Oracle(13, 7)
    _import_keys(o, hevm, ll)
    runner.lw.hevm_test_zero_encryption(hevm.vm, True)
    E, MULCC, MULCP, RS, BOOT = ha.OP_ENCODE, ha.OP_MULCC, ha.OP_MULCP, ha.OP_RESCALE, ha.OP_BOOTSTRAP
    ops = [(E, 0, 0xFFFF, (2 << 10) + 20),   # all-ones constant at scale 2^20 ("upscale")
           (MULCC, 1, 0, 0),                 # x^2: scale 2^80, 2 primes
           (MULCP, 2, 1, 0),                 # * 1 at 2^20: scale 2^100
           (RS, 3, 2, 0),                    # 1 prime, scale 2^100 / q_1 ~ 2^40      <- the ResNet program's opcode-10 operand
           (BOOT, 4, 3, 3),                  # re-encrypt at 3 primes, scale 2^40
           (BOOT, 5, 1, 5)]                  # and from scale 2^80 at 2 primes to 5 primes, scale 2^80
    hv = ha.pack_hevm([40], [2], [40, 40, 80, 80], [1, 3, 2, 5], [3, 4, 1, 5], 6, 1, 2, np.array(ops, dtype=np.uint16))
    hevm.load_mem(ha.pack_cst([]), hv)
    x = np.random.default_rng(5).uniform(-1, 1, o.slots)
    hevm.setInput(0, x)
    hevm.run()
    src40, out40, src80, out80 = (_get_ct(hevm, ll, r) for r in (3, 4, 1, 5))
    assert not out40.data[1].any() and not out80.data[1].any()      # c1 = 0: the hook is on, c0 is the plaintext itself
    for src, out, target, rel in ((src40, out40, 3, None), (src80, out80, 5, 2.0**-45)):
        new_scale = 2.0 ** int(np.log2(src.scale))                   # SEAL_HEVM.cpp:332 -> :262
        assert out.ell == target and out.scale == new_scale
        vals = o.decode(o.decrypt(src))                              # Decryptor::decrypt + CKKSEncoder::decode (real parts, :330-331)
        want = o.encode(vals, new_scale, target)                     # CKKSEncoder::encode at the target level (:332)
        d = _centered_diff(o, out.data[0], want.data, target)
        if rel is None:
            assert np.abs(d).max() <= 1, np.abs(d).max()             # the stated bound at scale 2^40
            assert (d != 0).mean() < 0.01                            # and almost every coefficient is identical
        else:
            assert np.abs(d).max() <= rel * new_scale, (np.abs(d).max(), rel * new_scale)
        # the re-encoded plaintext decodes to x^2 as well as the source did
        back = o.decode(Plaintext(out.data[0], new_scale))
        assert np.abs(back - x * x).max() < 1e-6
