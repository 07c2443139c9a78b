"""Shared helpers of the GPU tests: move key / plaintext / ciphertext limbs between a GPU VM and the oracle."""
import numpy as np

from oracle.oracle import Ciphertext, Oracle, OracleVM, Plaintext


def _import_keys(o: Oracle, hevm, ll):
    """pull the GPU VM's key material into the oracle so both interpret the program on identical limbs"""
    from dacapo_amd import runner

    lw = runner.lw
    K, N = o.K, o.N
    D = o.dnum if (o.ks, o.alpha) != (1, 1) else K - 1   # grouped digits: [dnum][2][K][N]
    o.sk = ll.read_device(lw.hevm_secret_key(hevm.vm), (K, N))
    o.pk = ll.read_device(lw.hevm_public_key(hevm.vm), (2, K, N))
    o.relin = ll.read_device(lw.hevm_relin_key(hevm.vm), (D, 2, K, N))
    o.galois = {}
    for elt in o.default_galois_elts():
        p = lw.hevm_galois_key(hevm.vm, elt)
        assert p, f"default Galois key {elt} missing"
        o.galois[elt] = ll.read_device(p, (D, 2, K, N))


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
        p = runner.lw.hevm_plain(hevm.vm, i, ctypes.byref(lvl), ctypes.byref(sc))
        if p:
            ovm.plains[i] = Plaintext(ll.read_device(p, (lvl.value, o.N)), sc.value)
    return ovm
