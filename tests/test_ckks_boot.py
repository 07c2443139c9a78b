"""CPU: real CKKS bootstrapping as an instruction sequence (dacapo_amd/ckks_boot.py; what `bootstrap` is in the reference's HEaaN
runtime, /root/reference/lib/Runtime/HEAAN_HEVM.cpp:386-399, and what opcode 10 only stands in for in its SEAL runtime).
  * the radix-2 factorisation of the special DFT for SEAL's slot order (generator 3) against the dense matrix;
  * the whole construction on cleartext slot vectors with the VM's exact scale semantics (ModRaise adds q0 * I);
  * the same program on REAL ciphertexts through the CPU oracle (its own NTT / key switch / rescale, plus the four extension
    opcodes): a ciphertext at 1 prime comes back at 3 primes, scale exactly 2^40, decrypting to the message."""
import numpy as np
import pytest

from dacapo_amd import ckks_boot as cb
from dacapo_amd import hevm_asm as ha


def _brev(i, bits):
    return int(format(i, f"0{bits}b")[::-1], 2) if bits else 0


@pytest.mark.parametrize("logN", [3, 4, 6])
def test_special_dft_factorisation_for_generator_3(logN):
    N, n = 1 << logN, 1 << (logN - 1)
    e = cb.slot_exponents(logN)
    zeta = np.exp(2j * np.pi / (2 * N))
    A0 = np.array([[zeta ** ((int(p) * j) % (2 * N)) for j in range(n)] for p in e])
    rng = np.random.default_rng(1)
    t = rng.normal(size=N)
    z = cb.embed(t, logN)
    assert np.abs(z - np.array([np.sum(t * zeta ** ((int(p) * np.arange(N)) % (2 * N))) for p in e])).max() < 1e-10
    P = [_brev(i, logN - 1) for i in range(n)]
    ms = [1 << i for i in range(1, logN)]
    m0 = rng.normal(size=n) + 0j
    x = m0[P].copy()
    for m in ms:                                             # SlotToCoeff direction: A0 m = S_n ... S_2 (m in bit-reversed order)
        x = cb.dft_factor(m, logN, herm=False).apply(x)
    assert np.abs(x - A0 @ m0).max() < 1e-10
    Dp = zeta ** ((e * n) % (2 * N))                         # zeta_k^(N/2) = +-i
    for half, pre in ((t[:n], np.ones(n)), (t[n:], np.conj(Dp))):  # CoeffToSlot: t = (2 / N) Re(A0^H [conj(D)] z)
        y = pre * z
        for m in reversed(ms):
            y = cb.dft_factor(m, logN, herm=True).apply(y)
        assert np.abs((2 / N) * y.real - half[P]).max() < 1e-10


def _boot_program(logN, r=5, target=3, in_level=1):
    K = target + cb.boot_levels(r) + 1
    b = ha.Builder(slots=1 << (logN - 1), init_level=in_level, shadow=False)
    x = b.input(None, level=in_level, scale_bits=40)
    em = cb.BootstrapEmitter(b, logN, K, target, r=r)
    y, label = em.bootstrap(x, 2.0**40)
    b.output(y)
    cst, hv, info = b.assemble()
    return em, cst, hv, info, label


def test_prime_chain_is_the_runtime_s():
    from oracle.oracle import Oracle

    assert cb.seal_prime_chain(12, 6) == Oracle(12, 6).primes     # CoeffModulus::Create order, special prime last


def test_bootstrap_on_cleartext_slots_with_exact_scale_semantics():
    logN = 11
    em, cst, hv, info, label = _boot_program(logN)
    assert label == 2.0**40
    mix = info["op_mix"]
    assert mix["mulcc"] == 32 and mix["rescale"] == 60 and mix["bootstrap"] == 0       # 2 x (1 + 10 + 5) squarings/products, no stand-in
    msg = np.random.default_rng(3).uniform(-1, 1, 1 << (logN - 1))
    outs, trace = cb.simulate(hv, cst, [msg], logN, em.primes, secret_weight=64, return_trace=True)
    assert np.abs(outs[0] - msg).max() < 1e-7 and np.abs(outs[0].imag).max() < 1e-7
    assert trace[-1][2] == 3 and trace[-1][3] == 2.0**40                                   # 3 primes left, label exactly 2^40
    assert max(t[2] for t in trace) == 20                                                  # ModRaise to the top of a 21-prime chain


def test_bootstrap_of_a_real_ciphertext_on_the_cpu_oracle(tmp_path):
    from oracle.oracle import Oracle, OracleVM

    logN = 10
    em, cst, hv, info, _ = _boot_program(logN)
    o = Oracle(logN, 21)
    assert o.primes == em.primes
    offs = sorted({(int(q) - 65536 if q >= 32768 else int(q)) for op, _, _, q in ha.unpack_hevm(hv)["ops"].tolist() if op == ha.OP_ROTATE} - {0})
    o.keygen_sparse(32, seed=3, galois_elts=sorted(set(o.default_galois_elts()) | {o.elt_from_step(s) for s in offs}))
    (tmp_path / "p.cst").write_bytes(cst)
    (tmp_path / "p.hevm").write_bytes(hv)
    vm = OracleVM(o)
    vm.load(tmp_path / "p.cst", tmp_path / "p.hevm")
    vm.preprocess()
    msg = np.random.default_rng(1).uniform(-1, 1, o.slots)
    vm.encrypt(0, msg)
    assert vm.ciphers[0].ell == 1
    vm.run()
    out_ct = vm.ciphers[vm.prog.res_dst[0]]
    assert out_ct.ell == 3 and out_ct.scale == 2.0**40
    assert np.abs(vm.decrypt_result(0) - msg).max() < 1e-6


def test_vm_scale_mirror_follows_the_reference_semantics():
    b = ha.Builder(slots=64, init_level=3, shadow=False)
    x = b.input(None, level=3, scale_bits=40)
    y = b.rescale(b.mul_plain(x, [0.5], scale_bits=60, normalise=False))
    primes = cb.seal_prime_chain(7, 4)
    sc = cb.vm_scales(b, primes)
    assert sc[y.id] == 2.0**40 * 2.0**60 / float(primes[2])                                # rescale divides by the dropped prime, as a double


def test_lowering_opcode_10_of_a_compiled_program_to_real_bootstrapping():
    """ckks_boot.lower_bootstraps: a program with opcode 10 (what the reference's compiler emits for `bootstrap`) re-emitted with every
    opcode 10 replaced by the real sequence; on cleartext slots (ModRaise overflow simulated) it still computes the same function"""
    logN, slots = 11, 1 << 10
    rng = np.random.default_rng(5)
    b = ha.Builder(slots=slots, init_level=3, policy="lazy", boot_level=3, shadow=True)
    x = b.input(rng.uniform(-1, 1, slots))
    y = x
    for _ in range(7):                                   # enough multiplications to run out of primes more than once
        y = b.add_plain(b.mul_plain(b.mul(y, y), [0.5]), [0.1])
    b.output(b.finish(b.add(y, b.rotate(x, 3))))
    cst, hv, info = b.assemble()
    assert info["op_mix"]["bootstrap"] >= 2
    hv2, cst2 = cb.lower_bootstraps(hv, cst, logN, 21, msg_bits=1)
    ops = ha.unpack_hevm(hv2)["ops"]
    assert int((ops[:, 0] == ha.OP_BOOTSTRAP).sum()) == 0 and int((ops[:, 0] == ha.OP_MODRAISE).sum()) == info["op_mix"]["bootstrap"]
    out = cb.simulate(hv2, cst2, [x.plain], logN, cb.seal_prime_chain(logN, 21))[0]
    assert np.abs(out.real - b.expected()[0]).max() < 1e-5 and np.abs(out.imag).max() < 1e-5


def _mixed_chain(logN, target=3, ks=1, r=5):
    """a HEaaN-style chain: 60-bit base prime, 51-bit rescale primes, 60-bit special primes (HEAAN_HEVM.cpp:55-56; profiled_HEAAN_GPU.json:
    rescalingFactor 51)"""
    K = target + cb.boot_levels(r) + ks
    return K, cb.mixed_prime_chain(logN, [60] + [51] * (K - 1 - ks) + [60] * ks)


def test_mixed_prime_chain_is_what_the_oracle_and_the_runtime_take():
    from oracle.oracle import Oracle

    K, primes = _mixed_chain(11)
    assert [int(q).bit_length() for q in primes] == [60] + [51] * (K - 2) + [60] and len(set(primes)) == K
    assert all(q % (2 << 11) == 1 and cb._is_prime(q) and (1 << q.bit_length()) - q < (1 << 28) for q in primes)
    assert Oracle(11, K, primes=primes).primes == primes
    assert cb.mixed_prime_chain(11, [60, 60, 60]) == cb.seal_prime_chain(11, 3)[::-1]     # same scan, CoeffModulus::Create lists it reversed


def test_bootstrap_on_a_mixed_60_51_bit_chain_cleartext_and_oracle(tmp_path):
    """round 4: the emitter's scale bookkeeping follows the chain it is given (every "60" became the width of the prime it meant): on a
    mixed chain EvalMod runs at scale ~2^51, the matrices are encoded at 2^51 / 2^46, ModRaise starts from the 60-bit base prime.  Cleartext
    slots with the VM's scale semantics: 1e-7; a real ciphertext through the CPU oracle (plain and grouped-digit keys): 1 prime -> 3 primes,
    label exactly 2^40, the message back within 2e-6 (20.4 - 20.7 bits measured: the 60-bit chain gives the same at this ring)."""
    from oracle.oracle import Oracle, OracleVM

    logN = 11
    K, primes = _mixed_chain(logN)
    b = ha.Builder(slots=1 << (logN - 1), init_level=1, shadow=False)
    x = b.input(None, level=1, scale_bits=40)
    em = cb.BootstrapEmitter(b, logN, K, 3, primes=primes)
    y, label = em.bootstrap(x, 2.0**40)
    b.output(y)
    cst, hv, info = b.assemble()
    assert label == 2.0**40 and em.diag_bits == 46 and em.boot_in_bits == 50
    msg = np.random.default_rng(3).uniform(-1, 1, 1 << (logN - 1))
    outs, trace = cb.simulate(hv, cst, [msg], logN, primes, secret_weight=64, return_trace=True)
    assert np.abs(outs[0] - msg).max() < 1e-7 and trace[-1][2] == 3 and trace[-1][3] == 2.0**40
    for ks in (1, 2):
        logN = 10
        K, primes = _mixed_chain(logN, ks=ks)
        K2, cst, hv, offs, em = cb.single_bootstrap_program(logN, ks=ks, primes=primes)
        assert K2 == K
        o = Oracle(logN, K, primes=primes)
        if ks > 1:
            o.set_hybrid(ks, ks)
        o.keygen_sparse(32, seed=3, galois_elts=sorted(set(o.default_galois_elts()) | {o.elt_from_step(s) for s in offs}))
        (tmp_path / "p.cst").write_bytes(cst)
        (tmp_path / "p.hevm").write_bytes(hv)
        vm = OracleVM(o)
        vm.load(tmp_path / "p.cst", tmp_path / "p.hevm")
        vm.preprocess()
        msg = np.random.default_rng(1).uniform(-1, 1, o.slots)
        vm.encrypt(0, msg)
        vm.run()
        out_ct = vm.ciphers[vm.prog.res_dst[0]]
        assert out_ct.ell == 3 and out_ct.scale == 2.0**40
        assert np.abs(vm.decrypt_result(0) - msg).max() < 2e-6, ks
