"""GPU: the modular arithmetic is no longer tied to 60-bit primes (round 3; modarith.hpp's width-tagged fold word, built as libSEAL_HEVM_gw.so beside the 60-bit-only default library): any chain of primes
q = 2^b - d, 45 <= b <= 60, d < 2^28 -- uniform 51-bit chains like the rescale primes of the reference's HEaaN configuration
(profiled_HEAAN_GPU.json: rescalingFactor 51), and MIXED chains (60-bit base and special primes around 51-bit rescale primes).  Against
the oracle (generic Barrett arithmetic on whatever primes it is given), limb for limb: NTT / inverse NTT, rotation hop, ct x ct +
relinearise, rescale -- at every level."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.oracle import Ciphertext, Oracle, splitmix_fill  # noqa: E402


def _chain(logN, widths):
    """one prime per entry of `widths`, CoeffModulus::Create style per width class (scan down from 2^b in steps of 2N), no repeats"""
    from dacapo_amd import ckks_boot as cb

    out, used = [], set()
    for b in widths:
        v = (1 << b) + 1
        while True:
            v -= 2 << logN
            if v not in used and cb._is_prime(v):
                used.add(v)
                out.append(v)
                break
    return out


@pytest.mark.parametrize("logN,widths", [(12, [51] * 5), (13, [48, 48, 48, 48]), (12, [60, 51, 51, 51, 60]), (12, [55, 60, 45, 51, 58, 60]),
                                         (12, [51] * 21), (12, [60] + [51] * 19 + [60])])  # (deep chains: more than 16 digits per inner product)
def test_ntt_and_evaluator_ops_on_other_prime_widths(logN, widths):
    from dacapo_amd import lowlevel as ll

    primes = _chain(logN, widths)
    K, N = len(primes), 1 << logN
    o = Oracle(logN, K, primes=primes)
    ctx = ll.Context(logN, primes=primes)
    assert ctx.primes == o.primes == primes and ctx.roots == [o.psi(p) for p in range(K)]
    L = ctx.L   # libSEAL_HEVM_gw.so: the generic-width build of the same sources
    assert L is ll.lib_gw()
    # NTT round trip and == oracle, every prime
    a = np.stack([splitmix_fill(17 + i, N) % np.uint64(primes[i]) for i in range(K)])
    a[0, : N // 2] = 0
    a[1, :] = np.uint64(primes[1] - 1)
    d = ll.DeviceBuffer.from_host(a)
    ctx.ntt(d, K)
    assert (d.to_host() == o.ntt_fwd(a, list(range(K)))).all()
    ctx.ntt(d, K, inverse=True)
    assert (d.to_host() == a).all()
    elt = o.elt_from_step(3)
    o.keygen(seed=5, galois_elts=[elt])
    dk, dr = ll.DeviceBuffer.from_host(o.galois[elt]), ll.DeviceBuffer.from_host(o.relin)
    for ell in range(1, K):
        q = np.array(primes[:ell], dtype=np.uint64)[:, None]
        x = np.stack([np.stack([splitmix_fill(1 + 7 * p + i + 100 * ell, N) for i in range(ell)]) % q for p in range(2)])
        y = np.stack([np.stack([splitmix_fill(99 + 7 * p + i + 100 * ell, N) for i in range(ell)]) % q for p in range(2)])
        dx, dy, dd = ll.DeviceBuffer.from_host(x), ll.DeviceBuffer.from_host(y), ll.DeviceBuffer((2, ell, N))
        st = ell * N
        X, Y = Ciphertext(x, 2.0**40), Ciphertext(y, 2.0**40)
        L.dc_ct_rotate_hop(ctx.h, dd.ptr, st, dx.ptr, st, elt, dk.ptr, ell, None)
        assert (dd.to_host() == o.apply_galois(X, elt).data).all(), ("rotate", ell)
        L.dc_ct_mul_relin(ctx.h, dd.ptr, st, dx.ptr, st, dy.ptr, st, dr.ptr, ell, None)
        want = o.mul_relin(X, Y)
        assert (dd.to_host() == want.data).all(), ("mul_relin", ell)
        if ell >= 2:
            L.dc_ct_rescale(ctx.h, dd.ptr, st, dd.ptr, st, ell, None)
            assert (dd.to_host()[:, : ell - 1] == o.rescale(want).data).all(), ("rescale", ell)


@pytest.mark.parametrize("fuse", [1, 2, 0])
@pytest.mark.parametrize("logN,widths,ks,alpha", [(12, [60, 51, 51, 51, 51, 51, 51, 60, 60], 2, 2), (12, [60] + [51] * 9 + [60] * 5, 5, 4),
                                                  (12, [58, 45, 51, 60, 48, 51, 60, 55], 3, 2)])
def test_grouped_digit_key_switch_on_mixed_chains(logN, widths, ks, alpha, fuse, request):
    """grouped-digit hybrid key switching on HEaaN-style MIXED chains -- a 60-bit base prime, 51-bit rescale primes, 60-bit special primes
    (HEAAN_HEVM.cpp:55-56, profiled_HEAAN_GPU.json: rescalingFactor 51) -- and on a deliberately ragged one: the base conversions move
    residues between width classes (a 60-bit residue into a 51-bit target and back), in all three launch sequences (fused with the conversions
    in the transforms' loaders / as matrix-core launches / round 3's).  Rotation hop and ct x ct + relinearise at every level == the oracle's
    orc_keyswitch_hybrid on the same primes, limb for limb.  (Round 3 computed garbage here without saying so: advisor finding 1.)"""
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    primes = _chain(logN, widths)
    K, N = len(primes), 1 << logN
    o = Oracle(logN, K, primes=primes)
    o.set_hybrid(ks, alpha)
    elt = o.elt_from_step(5)
    o.keygen(seed=11, galois_elts=[elt])
    ctx = ll.Context(logN, primes=primes, special=ks, alpha=alpha)
    L = ctx.L
    assert L is ll.lib_gw() and ctx.primes == primes and ctx.key_digits == o.dnum and ctx.max_level == o.max_level
    runner.set_option("hyb_fuse", fuse)
    request.addfinalizer(lambda: runner.set_option("hyb_fuse", 2))
    dk, dr = ll.DeviceBuffer.from_host(o.galois[elt]), ll.DeviceBuffer.from_host(o.relin)
    for ell in range(1, o.max_level + 1):
        q = np.array(primes[:ell], dtype=np.uint64)[:, None]
        x = np.stack([np.stack([splitmix_fill(1 + 7 * p + i + 100 * ell, N) for i in range(ell)]) % q for p in range(2)])
        y = np.stack([np.stack([splitmix_fill(99 + 7 * p + i + 100 * ell, N) for i in range(ell)]) % q for p in range(2)])
        dx, dy, dd = ll.DeviceBuffer.from_host(x), ll.DeviceBuffer.from_host(y), ll.DeviceBuffer((2, ell, N))
        st = ell * N
        X, Y = Ciphertext(x, 2.0**40), Ciphertext(y, 2.0**40)
        L.dc_ct_rotate_hop(ctx.h, dd.ptr, st, dx.ptr, st, elt, dk.ptr, ell, None)
        assert (dd.to_host() == o.apply_galois(X, elt).data).all(), ("rotate", ell)
        L.dc_ct_mul_relin(ctx.h, dd.ptr, st, dx.ptr, st, dy.ptr, st, dr.ptr, ell, None)
        assert (dd.to_host() == o.mul_relin(X, Y).data).all(), ("mul_relin", ell)


def test_both_builds_agree_on_the_reference_chain():
    """the generic-width build computes the 60-bit chain too (tag 0): same limbs as the default build, which keeps round 2's instruction
    streams (immediate shifts, no third fold)"""
    from dacapo_amd import lowlevel as ll

    ctx = ll.Context(13, 5)
    assert ctx.L is ll.lib() and all(p >> 59 == 1 for p in ctx.primes)
    gw = ll.lib_gw()
    arr = (__import__("ctypes").c_uint64 * 5)(*ctx.primes)
    h = gw.dc_context_create(13, 5, 60, arr)
    N = 1 << 13
    a = np.stack([splitmix_fill(3 + i, N) % np.uint64(ctx.primes[i]) for i in range(5)])
    d0, d1 = ll.DeviceBuffer.from_host(a), ll.DeviceBuffer.from_host(a)
    ctx.ntt(d0, 5)
    gw.dc_ntt_forward(h, d1.ptr, N, 5, None, 0, 0, None)
    assert (d0.to_host() == d1.to_host()).all()
    gw.dc_context_destroy(h)


def _vm_program(slots, levels, bits):
    from dacapo_amd import hevm_asm as ha

    rng = np.random.default_rng(6)
    # ciphertexts at 2^40, plaintexts at one prime's worth of scale (2^bits), one rescale per product: the lazy policy with `bits`-bit primes
    b = ha.Builder(slots=slots, init_level=levels, policy="lazy", boot_level=levels, rescale_bits=bits, shadow=True)
    x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
    t = b.add(b.mul(x, y), b.rotate(x, 3))
    t = b.add(b.mul_plain(t, rng.uniform(-1, 1, slots)), b.rotate(y, -5))
    u = b.mul(t, t)
    u = b.add(u, b.rotate(b.mul_plain(x, [0.25]), 33))
    b.output(b.finish(u))
    return b


def test_vm_program_on_a_51_bit_chain_matches_the_oracle_vm(tmp_path):
    """the whole boundary on 51-bit primes: key generation, encode, encrypt, a program with rotations (incl. multi-hop NAF offsets), ct x ct,
    ct x pt, rescales by 51-bit primes, decrypt / decode -- GPU VM (generic-width build, picked by runner.HEVM from option prime_bits) ==
    oracle VM limb for limb, in the same process as the 60-bit VMs of the other tests (round 3 needed a child: the binding was per process)"""
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner
    from gpu_helpers import _get_ct, _import_keys, _mirror_vm

    logN, K, bits = 12, 7, 51
    b = _vm_program(1 << (logN - 1), K - 1, bits)
    cst, hv, info = b.assemble()
    hevm = runner.HEVM(seed=77, logN=logN, num_primes=K, vm_options={"prime_bits": bits})
    o = Oracle(logN, K, bit_size=bits)
    assert [int(p).bit_length() for p in o.primes] == [bits] * K
    _import_keys(o, hevm, ll)
    # every key limb the GPU generated is a canonical residue of ITS prime (round 3's uniform sampler drew 60 bits whatever the width: on
    # narrow primes the keys' uniform halves came out non-canonical, and deep key switches then left the lazy accumulators' range)
    pr = np.array(o.primes, dtype=np.uint64)
    assert (o.sk < pr[:, None]).all() and (o.pk < pr[None, :, None]).all() and (o.relin < pr[None, None, :, None]).all()
    assert all((k < pr[None, None, :, None]).all() for k in o.galois.values())
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    for i, a in enumerate(b.args):
        hevm.setInput(i, a.plain)
        ovm.ciphers[i] = _get_ct(hevm, ll, i)
    hevm.run()
    ovm.run()
    r = ovm.prog.res_dst[0]
    got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
    assert got.ell == want.ell and got.scale == want.scale and (got.data == want.data).all()
    assert np.abs(hevm.getOutput()[0] - b.expected()[0]).max() < 1e-5
    assert info["op_mix"]["rescale"] >= 2 and info["op_mix"]["mulcc"] == 2
    hevm.close()


def test_lazy_sums_on_a_51_bit_grouped_digit_chain_match_the_oracle_vm(tmp_path):
    """option hyb_lazy_sum on the generic-width build (libSEAL_HEVM_gw.so): nine 51-bit primes, three of them special, digits of three -- sums of
    rotations divided by P once (hyb_mac_group_kernel, plan_exec.hip section 2b) == the oracle VM replaying the plan's groups on the same
    primes, limb for limb (tests/test_gpu_hybrid.py has the 60-bit build's version of this test, and what is eligible)"""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner
    from gpu_helpers import _get_ct, _import_keys, _mirror_vm

    logN, K, ks, bits = 12, 9, 3, 51
    slots = 1 << (logN - 1)
    rng = np.random.default_rng(31)
    b = ha.Builder(slots=slots, init_level=K - ks, policy="lazy", boot_level=K - ks, rescale_bits=bits, shadow=True)
    x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
    z = b.add(b.add(b.rotate(x, 4), y), b.rotate(y, -2))
    out = None
    for g in range(4):
        inner = b.add(b.mul_plain(x, rng.uniform(-1, 1, slots)), b.mul_plain(y, rng.uniform(-1, 1, slots)))
        if g:
            inner = b.rotate(inner, (8, 16, 40)[g - 1])          # 40 = 32 + 8: its second hop joins the sum
        out = inner if out is None else b.add(out, inner)
    ts = [b.mul_plain(x, [0.01 * (i + 1)]) for i in range(8)]   # (eight temporaries alive at once: no rotation result stays a register's FINAL
    pad = ts[0]                                                 # value -- architectural state, which would keep it out of its group)
    for t in ts[1:]:
        pad = b.add(pad, t)
    b.output(b.finish(out))
    b.output(b.finish(z))
    b.output(b.finish(pad))
    cst, hv, _ = b.assemble()
    hevm = runner.HEVM(seed=13, logN=logN, num_primes=K, ks_special=ks, vm_options={"prime_bits": bits, "hyb_lazy_sum": 1})
    assert hevm.lw is runner.bind_vm_lib(__import__("dacapo_amd").LIB_PATH_GW)
    o = Oracle(logN, K, bit_size=bits)
    o.set_hybrid(ks)
    _import_keys(o, hevm, ll)
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    for i, a in enumerate(b.args):
        hevm.setInput(i, a.plain)
        ovm.ciphers[i] = _get_ct(hevm, ll, i)
    hevm.run()
    groups = hevm.lazy_groups()
    assert sorted(len(g) for g in groups) == [2, 3], groups
    ovm.set_lazy_groups(groups)
    ovm.run()
    for k in range(2):
        r = ovm.prog.res_dst[k]
        got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
        assert got.ell == want.ell and got.scale == want.scale and (got.data == want.data).all(), k
        assert np.abs(hevm.getOutput()[k] - b.expected()[k]).max() < 1e-5
    hevm.close()
