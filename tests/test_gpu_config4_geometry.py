"""GPU: the grouped-digit key switch AT BASELINE CONFIG 4's GEOMETRY -- N = 2^17, 31 data + 8 special primes, digits of 7 primes, the base
conversions on the matrix cores (64 byte planes: the full width of the MFMA operand layout) -- against the oracle, limb for limb.  This is
the code that carries the config-4 run (dacapo_amd/csrc/hybrid_ks.hip; what it serves in the reference: HEAAN_HEVM.cpp:300-303 rotate,
:386-399 bootstrap), and tests/test_gpu_hybrid.py only reaches N <= 2^13, alpha <= 3:
  * kernel level (include/dacapo_ckks.h): one rotation hop and one ct x ct multiply + relinearise at levels 1, 3 (below the matrix-core
    threshold), 4 (at it), 7 (exactly one digit), 8 (a one-prime last digit), 14 (two digits: where the model runs), 31 (top: 5 digits,
    partial last one) == orc_rotate_ks_hybrid / orc_keyswitch_hybrid;
  * a hoisted batch through the VM's plan: five rotations of one ciphertext and two of another in one wave (shared decompositions), at the
    top level and again at 12 primes, direct keys and a two-hop offset mixed == the oracle VM, which rotates one instruction at a time;
  * the prefix of the nt = 2^16 ResNet trace on grouped-digit keys (ks = 8, alpha = 7): bit-identical to the oracle VM."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gpu_helpers import _get_ct, _import_keys, _mirror_vm  # noqa: E402
from oracle.oracle import Ciphertext, Oracle, lib as orc_lib, splitmix_fill  # noqa: E402

LOGN, KS, ALPHA = 17, 8, 7
K4 = 31 + KS
GOLDEN = Path(__file__).resolve().parent / "golden" / "resnet20_nt16"


def _threads():
    import os

    orc_lib().orc_set_threads(min(8, os.cpu_count() or 1))   # OpenMP over limbs: same arithmetic per limb (tests/test_oracle_hybrid.py)


# (8, 7): the shape config 4 has run on since round 3 (5 digits at the top level).  (9, 8): round 5's cheaper key shape -- digits of 8 under 9
# special primes, 4 digits, P still above a digit -- whose mod-down has 9 inputs: two K-chunks per matrix-core tile (hybrid_ks.hip).
@pytest.mark.parametrize("KS,ALPHA", [(8, 7), (9, 8)])
def test_rotate_hop_and_mul_relin_at_config4_geometry_match_the_oracle(KS, ALPHA):
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    _threads()
    N = 1 << LOGN
    K4 = 31 + KS
    o = Oracle(LOGN, K4)
    o.set_hybrid(KS, ALPHA)
    ctx = ll.Context(LOGN, K4, special=KS, alpha=ALPHA)
    assert ctx.primes == o.primes and ctx.key_digits == o.dnum == -(-31 // ALPHA) and ctx.max_level == o.max_level == 31
    pr = np.array(o.primes, dtype=np.uint64)
    # a key is a constant of the key switch: uniform limbs exercise every residue path (a real key's limbs ARE uniform)
    keys = []
    for s in (7, 1007):
        k = np.stack([splitmix_fill(s + i, N) for i in range(o.dnum * 2 * K4)]).reshape(o.dnum, 2, K4, N)
        k %= pr[None, None, :, None]
        keys.append(k)
    elt = o.elt_from_step(5)
    o.galois[elt], o.relin = keys
    dk, dr = ll.DeviceBuffer.from_host(keys[0]), ll.DeviceBuffer.from_host(keys[1])
    L = ll.lib()
    for ell in ((1, 3, 4, 7, 8, 14, 31) if ALPHA == 7 else (3, 4, 8, 9, 14, 31)):  # (alpha = 8: one digit exactly at 8, a one-prime last digit at 9)
        q = pr[:ell, None]
        a = np.stack([np.stack([splitmix_fill(1 + 7 * p + i + 100 * ell, N) for i in range(ell)]) % q for p in range(2)])
        b = np.stack([np.stack([splitmix_fill(99 + 7 * p + i + 100 * ell, N) for i in range(ell)]) % q for p in range(2)])
        da, db, dd = ll.DeviceBuffer.from_host(a), ll.DeviceBuffer.from_host(b), ll.DeviceBuffer((2, ell, N))
        st = ell * N
        A, B = Ciphertext(a, 2.0**40), Ciphertext(b, 2.0**40)
        want_rot, want_mul = o.apply_galois(A, elt).data, o.mul_relin(A, B).data
        # hyb_fuse 1: the fused sequence, conversions in the transforms' loaders (hybrid_fused.hip); 2: the fused sequence with the conversions
        # as separate matrix-core launches; 0: round 3's sequence (hybrid_ks.hip).  One oracle result, three GPU implementations.
        for fuse in (1, 2, 0):
            with runner.options(hyb_fuse=fuse):
                L.dc_memset(dd.ptr, 0xFF, dd.nbytes)
                L.dc_ct_rotate_hop(ctx.h, dd.ptr, st, da.ptr, st, elt, dk.ptr, ell, None)
                assert (dd.to_host() == want_rot).all(), ("rotate", ell, fuse)
                L.dc_ct_mul_relin(ctx.h, dd.ptr, st, da.ptr, st, db.ptr, st, dr.ptr, ell, None)
                assert (dd.to_host() == want_mul).all(), ("mul_relin", ell, fuse)


@pytest.mark.parametrize("plan,fuse,KS,ALPHA", [(1, 1, 8, 7), (0, 1, 8, 7), (1, 2, 8, 7), (1, 0, 8, 7), (1, 2, 9, 8), (0, 2, 9, 8)])
def test_hoisted_rotation_batch_at_config4_geometry_matches_the_oracle_vm(tmp_path, plan, fuse, KS, ALPHA):
    K4 = 31 + KS
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    _threads()
    slots = 1 << (LOGN - 1)
    rng = np.random.default_rng(17)
    b = ha.Builder(slots=slots, init_level=31, policy="lazy", boot_level=31, shadow=True)
    x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
    offs_x, offs_y = (1, 2, 3, 5, 8), (7, 6)                           # 6 = 8 - 2 under the default keys: two hops; the others get direct keys
    direct = [1, 2, 3, 5, 8, 7]

    def taps(u, v):
        acc = None
        for k, w in zip(offs_x, rng.uniform(-1, 1, len(offs_x))):
            t = b.mul_plain(b.rotate(u, k), [float(w)])
            acc = t if acc is None else b.add(acc, t)
        for k in offs_y:
            acc = b.add(acc, b.mul_plain(b.rotate(v, k), [0.5]))
        return acc

    top = taps(x, y)                                                   # seven rotations in one wave at 31 primes
    xl, yl = b.modswitch(x, 19), b.modswitch(y, 19)                    # ... and at 12 primes, where config 4's convolutions run
    low = taps(xl, yl)
    b.output(b.finish(top))
    b.output(b.finish(low))
    cst, hv, _ = b.assemble()
    runner.set_option("hyb_fuse", fuse)                                # (a launch-shape option: read at every launch, i.e. when the plan's graph is recorded)
    hevm = runner.HEVM(seed=5, logN=LOGN, num_primes=K4, ks_special=KS, ks_alpha=ALPHA, vm_options={"plan": plan})
    assert hevm.max_level == 31 and hevm.key_digits == -(-31 // ALPHA)
    hevm.addRotationKeys(direct)
    o = Oracle(LOGN, K4)
    o.set_hybrid(KS, ALPHA)
    elts = sorted({o.elt_from_step(s) for s in direct + [8, -2]})
    _import_keys(o, hevm, ll, elts=elts, relin=False)
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    for i, a in enumerate(b.args):
        hevm.setInput(i, a.plain)
        ovm.ciphers[i] = _get_ct(hevm, ll, i)
    hevm.run()
    ovm.run()
    for i in range(2):
        r = ovm.prog.res_dst[i]
        got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
        assert got.ell == want.ell and got.scale == want.scale
        assert (got.data == want.data).all(), (plan, i)
    out = hevm.getOutput()
    for i in range(2):                                                 # (switching noise at N = 2^17 with 7-prime digits: 4e-5 measured)
        assert np.abs(out[i] - b.expected()[i]).max() < 5e-4
    hevm.close()
    runner.set_option("hyb_fuse", 2)


@pytest.mark.parametrize("mode", [1, 2])
def test_lazy_sums_at_config4_geometry_match_the_oracle_vm(tmp_path, mode):
    """option hyb_lazy_sum at config 4's shape (N = 2^17, digits of 8 under 9 special primes): the giant steps of a BSGS product -- three
    rotations of three different inner sums added to a fourth -- at 31 primes (4 digits) and at 12 (2 digits, a partial one), one division
    by P per sum (hybrid_fused.hip hybf_rotate_sum) == the oracle VM replaying the plan's groups (orc_rotate_acc_hybrid / orc_moddown_hybrid)"""
    KS, ALPHA = 9, 8
    K4 = 31 + KS
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    _threads()
    slots = 1 << (LOGN - 1)
    rng = np.random.default_rng(23)
    b = ha.Builder(slots=slots, init_level=31, policy="lazy", boot_level=31, shadow=True)
    x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
    giants = (16, 32, 48)                                              # 48 gets a direct key

    def bsgs(u, v):
        out = b.add(b.mul_plain(u, [0.5]), b.mul_plain(v, [0.25]))
        for g, k in enumerate(giants):
            inner = b.add(b.mul_plain(u, [0.1 * (g + 1)]), b.mul_plain(v, [-0.2 * (g + 1)]))
            out = b.add(out, b.rotate(inner, k))
        return out

    top = bsgs(x, y)
    low = bsgs(b.modswitch(x, 19), b.modswitch(y, 19))
    # (eight temporaries alive at once: every register a rotation result sat in is written again, so none of them is a register's FINAL value --
    # architectural state the plan keeps observable, which would keep that rotation out of its group)
    ts = [b.mul_plain(x, [0.01 * (i + 1)]) for i in range(8)]
    pad = ts[0]
    for t in ts[1:]:
        pad = b.add(pad, t)
    b.output(b.finish(top))
    b.output(b.finish(low))
    b.output(b.finish(pad))
    cst, hv, _ = b.assemble()
    hevm = runner.HEVM(seed=5, logN=LOGN, num_primes=K4, ks_special=KS, ks_alpha=ALPHA, vm_options={"plan": 1, "hyb_lazy_sum": 1})
    hevm.addRotationKeys([48])
    o = Oracle(LOGN, K4)
    o.set_hybrid(KS, ALPHA)
    _import_keys(o, hevm, ll, elts=sorted({o.elt_from_step(s) for s in giants}), relin=False)
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    for i, a in enumerate(b.args):
        hevm.setInput(i, a.plain)
        ovm.ciphers[i] = _get_ct(hevm, ll, i)
    with runner.options(hyb_lazy_sum=mode):                            # 1: hyb_mac_group_kernel; 2: per-item accumulators + hybf_group_sum_kernel
        hevm.run()
    groups = hevm.lazy_groups()
    assert [len(g) for g in groups] == [3, 3], groups
    ovm.set_lazy_groups(groups)
    ovm.run()
    for i in range(2):
        r = ovm.prog.res_dst[i]
        got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
        assert got.ell == want.ell and got.scale == want.scale
        assert (got.data == want.data).all(), i
    out = hevm.getOutput()
    for i in range(2):
        assert np.abs(out[i] - b.expected()[i]).max() < 5e-4
    hevm.close()


def test_double_hoisting_at_config4_geometry_matches_the_oracle_vm(tmp_path):
    """option hyb_double_hoist at config 4's shape (N = 2^17, digits of 8 under 9 special primes): a convolution's taps -- three rotations, each
    times its own plaintext, added to an unrotated tap -- at 31 primes (4 digits) and at 12 (2 digits, a partial one): the products are taken in
    the raised basis on the plaintexts' special-prime limbs (encoded by the plan, hevm_plain_special), one division by P per sum == the oracle VM
    replaying the plan's groups with Oracle.lazy_mul_plain on the same plaintext limbs"""
    KS, ALPHA = 9, 8
    K4 = 31 + KS
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    _threads()
    slots = 1 << (LOGN - 1)
    rng = np.random.default_rng(29)
    b = ha.Builder(slots=slots, init_level=31, policy="lazy", boot_level=31, shadow=True)
    x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
    taps = (16, 32, 48)                                                # 48 gets a direct key

    def conv(u, v):
        out = b.mul_plain(u, rng.uniform(-1, 1, slots))
        for k, src in zip(taps, (u, v, u)):
            out = b.add(out, b.mul_plain(b.rotate(src, k), rng.uniform(-1, 1, slots)))
        return out

    top = conv(x, y)
    low = conv(b.modswitch(x, 19), b.modswitch(y, 19))
    ts = [b.mul_plain(x, [0.01 * (i + 1)]) for i in range(8)]          # (no rotation result is a register's final value: see the test above)
    pad = ts[0]
    for t in ts[1:]:
        pad = b.add(pad, t)
    b.output(b.finish(top))
    b.output(b.finish(low))
    b.output(b.finish(pad))
    cst, hv, _ = b.assemble()
    hevm = runner.HEVM(seed=5, logN=LOGN, num_primes=K4, ks_special=KS, ks_alpha=ALPHA, vm_options={"plan": 1, "hyb_lazy_sum": 1, "hyb_double_hoist": 1})
    hevm.addRotationKeys([48])
    o = Oracle(LOGN, K4)
    o.set_hybrid(KS, ALPHA)
    _import_keys(o, hevm, ll, elts=sorted({o.elt_from_step(s) for s in taps}), relin=False)
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    assert len(ovm.plains_special) == 6                                # the six taps' plaintexts, nothing else
    for i, a in enumerate(b.args):
        hevm.setInput(i, a.plain)
        ovm.ciphers[i] = _get_ct(hevm, ll, i)
    hevm.run()
    groups = hevm.lazy_groups()
    assert [len(g) for g in groups] == [3, 3], groups
    ovm.set_lazy_groups(groups)
    ovm.run()
    for i in range(2):
        r = ovm.prog.res_dst[i]
        got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
        assert got.ell == want.ell and got.scale == want.scale
        assert (got.data == want.data).all(), i
    out = hevm.getOutput()
    for i in range(2):
        assert np.abs(out[i] - b.expected()[i]).max() < 5e-4
    hevm.close()


def test_nt16_prefix_bit_exact_at_n17_on_grouped_digit_keys(tmp_path):
    """tests/test_gpu_config4.py::test_nt16_prefix_bit_exact_at_n17 with ks_special = 8, ks_alpha = 7: the stem convolution of the nt = 2^16
    trace as config 4 runs it (the .b14 lowering: 27 rotations at 14 primes as NAF hops of the default Galois keys, 25 ct x pt, 2 rescales) through the grouped-digit sequence, rotations of one
    source sharing their decomposition"""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    _threads()
    import gzip

    fx = ha.read_fixture(GOLDEN)
    hv0 = gzip.open(str(GOLDEN) + ".b14.hevm.gz").read()               # the lowering config 4 runs: inputs at 14 primes -> two digits
    prog = ha.unpack_hevm(hv0)
    ops, init_level = prog["ops"], int(prog["init_level"])
    first_boot = int(np.nonzero(ops[:, 0] == ha.OP_BOOTSTRAP)[0][0])
    hv, lvl, _ = ha.truncate_hevm(hv0, first_boot)
    K = init_level + KS
    hevm = runner.HEVM(seed=0x4845564D, logN=LOGN, num_primes=K, ks_special=KS, ks_alpha=ALPHA)
    assert hevm.max_level == init_level == 14 and hevm.key_digits == 2
    hevm.load_mem(fx["cst"], hv)
    o = Oracle(LOGN, K)
    o.set_hybrid(KS, ALPHA)
    rot = {int(np.array(r, dtype=np.uint16).astype(np.int16)) for op, _, _, r in ops[:first_boot].tolist() if op == ha.OP_ROTATE}
    o.galois = dict.fromkeys(o.default_galois_elts())                   # the default-set hops this prefix takes (Evaluator::rotate_internal's NAF)
    elts = {e for s in rot for e in o.rotate_hops(s)}
    _import_keys(o, hevm, ll, elts=sorted(elts), relin=False)
    ovm = _mirror_vm(hevm, ll, o, fx["cst"], hv, tmp_path)
    hevm.setInput(0, fx["packed"])
    ovm.ciphers[0] = _get_ct(hevm, ll, 0)
    hevm.run()
    ovm.run()
    reg = ovm.prog.res_dst[0]
    got, want = _get_ct(hevm, ll, reg), ovm.ciphers[reg]
    assert got.ell == want.ell == lvl and got.scale == want.scale
    assert (got.data == want.data).all()
    assert hevm.stats()["keyswitches"] >= 27
    hevm.close()
