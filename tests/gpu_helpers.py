"""Shared helpers of the GPU tests: move key / plaintext / ciphertext limbs between a GPU VM and the oracle."""
import numpy as np

from oracle.oracle import Ciphertext, Oracle, OracleVM, Plaintext


def _import_keys(o: Oracle, hevm, ll, elts=None, relin=True):
    """pull the GPU VM's key material into the oracle so both interpret the program on identical limbs.  `elts`: the Galois elements to
    import (default: SEAL's default set); at N = 2^17 / 39 primes a key is 0.4 GB, so the big-geometry tests name the ones they use."""
    lw = hevm.lw   # (the build of the library this VM lives in: the generic-width one for chains with narrow primes)
    K, N = o.K, o.N
    D = o.dnum if (o.ks, o.alpha) != (1, 1) else K - 1   # grouped digits: [dnum][2][K][N]
    o.sk = ll.read_device(lw.hevm_secret_key(hevm.vm), (K, N))
    o.pk = ll.read_device(lw.hevm_public_key(hevm.vm), (2, K, N))
    o.relin = ll.read_device(lw.hevm_relin_key(hevm.vm), (D, 2, K, N)) if relin else None
    o.galois = {}
    for elt in (o.default_galois_elts() if elts is None else elts):
        p = lw.hevm_galois_key(hevm.vm, elt)
        assert p, f"Galois key {elt} missing"
        o.galois[elt] = ll.read_device(p, (D, 2, K, N))


def run_conv_shaped_program(logN, K, tmp_path, seed=31):
    """a convolution-shaped program -- 19 rotated ct x pt products + 9 bare rotated ciphertexts summed (more than one 16-term and one 8-term
    reduction window of the n-ary sum), a second reader that is a rotation (the sum is materialised), a ct x ct product and rescales at
    several levels, 29 rotations incl. multi-hop ones -- on the GPU VM and on the oracle VM: returns what a test asserts on.  Launch-shape
    options in force around the call select the kernel forms (the plan's graph is recorded by load_mem)."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    slots = 1 << (logN - 1)
    rng = np.random.default_rng(9)
    b = ha.Builder(slots=slots, init_level=K - 1, policy="lazy", boot_level=K - 1, shadow=True)
    x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
    acc = None
    for k in range(19):
        t = b.mul_plain(b.rotate(x if k % 3 else y, 1 << (k % 7)), rng.uniform(-1, 1, slots))
        acc = t if acc is None else b.add(acc, t)
    z = b.mul_plain(y, rng.uniform(-1, 1, slots))
    for k in range(9):
        acc = b.add(acc, b.rotate(z, 3 + 2 * k))
    u = b.add(acc, b.rotate(acc, 5))
    b.output(b.finish(b.mul(u, u)))
    cst, hv, info = b.assemble()
    hevm = runner.HEVM(seed=seed, logN=logN, num_primes=K)
    o = Oracle(logN, K)
    _import_keys(o, hevm, ll)
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    for i, a in enumerate(b.args):
        hevm.setInput(i, a.plain)
        ovm.ciphers[i] = _get_ct(hevm, ll, i)
    hevm.run()
    ovm.run()
    r = ovm.prog.res_dst[0]
    got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
    out = hevm.getOutput()[0]
    res = {"limbs_identical": bool(got.ell == want.ell and (got.data == want.data).all()), "scale_identical": bool(got.scale == want.scale),
           "max_error_vs_cleartext": float(np.abs(out - b.expected()[0]).max()), "op_mix": info["op_mix"], "stats": hevm.stats()}
    hevm.close()
    return res


def run_multi_output_conv_program(logN, K, tmp_path, seed=37, outputs=11):
    """a convolution with several output channels: 12 rotated inputs (6 offsets of two ciphertexts), every output channel a sum over them with
    its own plaintexts -- some channels skip taps, one names a tap twice, some add a bare rotated ciphertext -- and every channel read by a
    rotation and by an addition afterwards (two readers: the sums are materialised, not folded into a rescale), on the GPU VM and on the oracle VM.  The sums of the channels form one step of the
    plan and share their sources: what batch_ops.hip b_sum_group_kernel is for."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    slots = 1 << (logN - 1)
    rng = np.random.default_rng(11)
    b = ha.Builder(slots=slots, init_level=K - 1, policy="lazy", boot_level=K - 1, shadow=True)
    x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
    taps = [b.rotate(x if t % 2 else y, 1 + 3 * (t // 2)) for t in range(12)]
    bare = b.mul_plain(x, rng.uniform(-1, 1, slots))  # a value at the products' scale, added as it is
    chans = []
    for c in range(outputs):
        acc = None
        for t, r in enumerate(taps):
            if (c * 5 + t) % 7 == 0 and c % 3 == 1:
                continue  # this channel skips the tap
            term = b.mul_plain(r, rng.uniform(-1, 1, slots))
            acc = term if acc is None else b.add(acc, term)
            if c == 4 and t == 3:  # the same tap again, another plaintext
                acc = b.add(acc, b.mul_plain(r, rng.uniform(-1, 1, slots)))
        if c % 4 == 2:
            acc = b.add(acc, bare)
        chans.append(acc)
    total, extra = None, None
    for c, ch in enumerate(chans):  # two readers per channel: a rotation (behind the lazy policy's rescale) and a plain addition
        r = b.rotate(ch, 2 + c)
        total = r if total is None else b.add(total, r)
        extra = ch if extra is None else b.add(extra, ch)
    b.output(b.finish(b.add(total, b.rescale(extra))))
    cst, hv, info = b.assemble()
    hevm = runner.HEVM(seed=seed, logN=logN, num_primes=K)
    o = Oracle(logN, K)
    _import_keys(o, hevm, ll)
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    for i, a in enumerate(b.args):
        hevm.setInput(i, a.plain)
        ovm.ciphers[i] = _get_ct(hevm, ll, i)
    hevm.run()
    ovm.run()
    r = ovm.prog.res_dst[0]
    got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
    out = hevm.getOutput()[0]
    res = {"limbs_identical": bool(got.ell == want.ell and (got.data == want.data).all()), "scale_identical": bool(got.scale == want.scale),
           "max_error_vs_cleartext": float(np.abs(out - b.expected()[0]).max()), "op_mix": info["op_mix"], "stats": hevm.stats()}
    hevm.close()
    return res


def _get_ct(hevm, ll, reg):
    c = hevm.getCtxt(reg)
    full = ll.read_device(c.data, (2, c.poly_stride // hevm.N, hevm.N))
    return Ciphertext(np.ascontiguousarray(full[:, : c.level]), c.scale)


def _mirror_vm(hevm, ll, o, cst, hv, tmp_path):
    import ctypes

    from dacapo_amd import runner

    (tmp_path / "p.cst").write_bytes(cst)
    (tmp_path / "p.hevm").write_bytes(hv)
    ovm = OracleVM(o)
    ovm.load(tmp_path / "p.cst", tmp_path / "p.hevm")
    for i in range(ovm.prog.num_ptxt):
        lvl, sc = ctypes.c_int32(), ctypes.c_double()
        p = hevm.lw.hevm_plain(hevm.vm, i, ctypes.byref(lvl), ctypes.byref(sc))
        if p:
            ovm.plains[i] = Plaintext(ll.read_device(p, (lvl.value, o.N)), sc.value)
    # option hyb_double_hoist: the special-prime limbs the plan encoded for the plaintexts that multiply rotations inside its lazy sums
    ovm.plains_special = {}
    if hasattr(hevm.lw, "hevm_plain_special") and getattr(o, "ks", 1) > 1:
        for i in range(ovm.prog.num_ptxt):
            p = hevm.lw.hevm_plain_special(hevm.vm, i)
            if p:
                ovm.plains_special[i] = ll.read_device(p, (o.ks, o.N))
    return ovm
