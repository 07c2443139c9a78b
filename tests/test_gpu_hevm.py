"""GPU: the HEVM drop-in boundary (include/hevm_abi.h) end to end -- runner-style call sequence on assembled
programs, checked (a) bit-exactly against the oracle VM fed with the same key/plaintext/input limbs and
(b) numerically (RMS) against the plaintext computation, as examples/tests/*.py do."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.oracle import Ciphertext, Oracle, OracleVM, Plaintext, read_cst, read_hevm


from gpu_helpers import _get_ct, _import_keys, _mirror_vm  # noqa: E402


@pytest.fixture(scope="module", params=["plan", "eager", "plan1", "dag"])
def vm13(request):
    """plan   = default: the batched execution plan (SSA-renamed registers, one batched launch sequence per wave; independent
             steps of a wave on an auxiliary stream; the whole sequence replayed as one HIP graph);
    eager  = the reference's dispatch loop, one instruction at a time on one stream (option plan = 0);
    plan1  = the plan issued launch by launch on one stream (no graph, no auxiliary stream);
    dag    = the plan's graph BUILT from its own dependencies (option plan_graph = 2: per-step single-stream captures copied into one
             graph with explicit edges, no per-wave fork / join) instead of captured from two streams."""
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    opts = {"plan": {}, "eager": {"plan": 0}, "plan1": {"plan_lanes": 1, "plan_graph": 0}, "dag": {"plan_graph": 2}}[request.param]
    hevm = runner.HEVM(seed=0x4845564D, logN=13, num_primes=7, vm_options=opts)
    hevm.mode = request.param
    o = Oracle(13, 7)
    _import_keys(o, hevm, ll)
    return hevm, o, ll


def test_generated_keys_satisfy_key_equations(vm13):
    hevm, o, ll = vm13
    K, N = o.K, o.N
    # secret is ternary; pk decrypts to small noise
    s = o.ntt_inv(o.sk, list(range(K)))
    for i, q in enumerate(o.primes):
        assert set(np.unique(s[i]).tolist()) <= {0, 1, q - 1}
    e = o.ntt_inv(o.poly_add(o.pk[0], o.poly_mul(o.pk[1], o.sk)), list(range(K)))
    cent = np.where(e[0] > o.primes[0] // 2, e[0].astype(np.int64) - np.int64(o.primes[0]), e[0].astype(np.int64))
    assert np.abs(cent).max() <= 21 and cent.std() > 2.0
    # relin digit j: c0 + c1 s - [i==j] (P mod q_j) s^2 is small
    sk2 = o.poly_mul(o.sk, o.sk)
    for j in (0, K - 2):
        d = o.poly_add(o.relin[j, 0], o.poly_mul(o.relin[j, 1], o.sk))
        fac = np.zeros((K, N), dtype=np.uint64)
        fac[j, :] = o.primes[K - 1] % o.primes[j]
        d = o.poly_sub(d, o.poly_mul(sk2, fac))
        e = o.ntt_inv(d, list(range(K)))
        for i in (0, j, K - 1):
            q = o.primes[i]
            c = np.where(e[i] > q // 2, e[i].astype(np.int64) - np.int64(q), e[i].astype(np.int64))
            assert np.abs(c).max() <= 21


def test_sobel_program_bit_exact_and_rms(vm13, tmp_path):
    from dacapo_amd import hevm_asm as ha

    hevm, o, ll = vm13
    rng = np.random.default_rng(100)
    img = rng.uniform(0, 1, 4096)
    b = ha.sobel_filter(img, slots=o.slots, init_level=6)
    cst, hv, info = b.assemble()
    hevm.load_mem(cst, hv)
    assert hevm.arglen == 1 and hevm.reslen == 1
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    # encoder parity: GPU-VM plaintext limbs vs oracle encode of the same constants (FP contraction may differ by 1 ulp)
    ovm2 = OracleVM(o)
    ovm2.load(tmp_path / "p.cst", tmp_path / "p.hevm")
    ovm2.preprocess()
    for i, (a, bb) in enumerate(zip(ovm.plains, ovm2.plains)):
        ca = o.ntt_inv(a.data, list(range(a.ell))).astype(np.int64)
        cb = o.ntt_inv(bb.data, list(range(bb.ell))).astype(np.int64)
        assert a.scale == bb.scale and np.abs(ca - cb).max() <= 1
    hevm.setInput(0, img)
    ovm.ciphers[0] = _get_ct(hevm, ll, 0)
    assert ovm.ciphers[0].ell == 6 and ovm.ciphers[0].scale == 2.0**40
    # fresh encryption decrypts correctly under the oracle too
    assert np.abs(o.decode(o.decrypt(ovm.ciphers[0])) - img[np.arange(o.slots) % 4096]).max() < 1e-6
    hevm.run()
    ovm.run()
    got = _get_ct(hevm, ll, ovm.prog.res_dst[0])
    want = ovm.ciphers[ovm.prog.res_dst[0]]
    assert got.ell == want.ell and got.scale == want.scale
    assert (got.data == want.data).all()  # program-level bit-exactness
    res = hevm.getOutput()
    # a second run() on the same inputs (a graph replay in plan mode) reproduces the same limbs
    hevm.run()
    again = _get_ct(hevm, ll, ovm.prog.res_dst[0])
    assert (again.data == want.data).all() and again.scale == want.scale and again.ell == want.ell
    ref = b.expected()[0]
    rms = np.sqrt(np.mean((res[0] - ref) ** 2))
    assert rms < 1e-4 * max(1.0, np.abs(ref).max()), rms
    assert np.abs(ovm.decrypt_result(0) - res[0]).max() < 1e-9  # host decoder == oracle decoder
    st = hevm.stats()
    assert st["op_counts"][1] == 9 and st["op_counts"][8] == 4 and st["keyswitches"] >= 4 + 8


def test_linear_regression_two_inputs_two_outputs(vm13):
    from dacapo_amd import hevm_asm as ha

    hevm, o, ll = vm13
    rng = np.random.default_rng(100)
    xs = rng.uniform(-1, 1, 4096)
    ys = 0.7 * xs + 0.2 + rng.normal(0, 0.01, 4096)
    b = ha.linear_regression(xs, ys, epochs=2, logn_data=12, slots=o.slots, init_level=6, min_level=1)
    cst, hv, info = b.assemble()
    hevm.load_mem(cst, hv)
    assert hevm.arglen == 2 and hevm.reslen == 2
    hevm.setInput(0, xs)
    hevm.setInput(1, ys)
    hevm.run()
    res = hevm.getOutput()
    for r, e in zip(res, b.expected()):
        assert np.sqrt(np.mean((r - e) ** 2)) < 1e-4


def test_rotation_by_arbitrary_offsets_and_bootstrap_opcode(vm13, tmp_path):
    from dacapo_amd import hevm_asm as ha

    hevm, o, ll = vm13
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, o.slots)
    b = ha.Builder(slots=o.slots, init_level=6)
    v = b.input(x)
    acc = b.rotate(v, 37)          # NAF: 3 hops
    acc = b.add(acc, b.rotate(v, -100))
    acc = b.add(acc, b.rotate(v, 4))
    m = b.mul(acc, acc)            # scale 80
    m = b.mul_plain(m, [0.5])      # 120 -> rescale
    z = b.bootstrap(m, 5)          # SEAL VM's decrypt/re-encrypt
    z = b.add_plain(z, [0.25])
    b.output(z)
    cst, hv, info = b.assemble()
    hevm.load_mem(cst, hv)
    hevm.setInput(0, x)
    hevm.run()
    res = hevm.getOutput()[0]
    assert np.sqrt(np.mean((res - b.expected()[0]) ** 2)) < 1e-4
    c = hevm.getCtxt(int(read_hevm_bytes(hv).res_dst[0]))
    assert c.level == 5


def read_hevm_bytes(hv):
    import tempfile

    with tempfile.NamedTemporaryFile(suffix=".hevm") as f:
        f.write(hv)
        f.flush()
        return read_hevm(f.name)


def test_file_based_runner_sequence(tmp_path):
    """the reference's call sequence through files: create_context -> initFullVM -> load -> encrypt -> run -> decrypt"""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import runner

    keydir = tmp_path / "keys"
    hevm = runner.HEVM(path=str(keydir), vm_options={"logn": 12, "primes": 4})
    assert sorted(p.name for p in keydir.iterdir()) == ["gal.seal", "parm.seal", "pub.seal", "relin.seal", "sec.seal"]
    assert hevm.logN == 12 and hevm.K == 4
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, hevm.slots)
    b = ha.Builder(slots=hevm.slots, init_level=3)
    v = b.input(x)
    y = b.add(b.mul(v, b.rotate(v, 1)), b.mul_plain(v, [2.0]))
    b.output(y)
    info = b.write(tmp_path / "_hecate_t.cst", tmp_path / "t.40._hecate_t.hevm")
    hevm.load(str(tmp_path / "_hecate_t.cst"), str(tmp_path / "t.40._hecate_t.hevm"))
    hevm.setInput(0, x)
    hevm.run()
    res = hevm.getOutput()[0]
    assert np.sqrt(np.mean((res - b.expected()[0]) ** 2)) < 1e-4
    # a second VM instance loading the same key directory decrypts what the first one produced? (server/client split)
    client = runner.HEVM(path=str(keydir), option="client")
    client.load(str(tmp_path / "_hecate_t.cst"), str(tmp_path / "t.40._hecate_t.hevm"))
    assert client.arglen == 1 and client.reslen == 1


def test_throughput_mode_three_streams_bit_exact(tmp_path):
    """hevm_set_streams: 3 independent ciphertext streams through one batched plan; every stream's result limbs equal
    the oracle VM's on that stream's input, and decrypt to that stream's expected plaintext."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D, logN=13, num_primes=7)
    o = Oracle(13, 7)
    _import_keys(o, hevm, ll)
    hevm.set_streams(3)
    rng = np.random.default_rng(11)
    imgs = [rng.uniform(0, 1, 4096) for _ in range(3)]
    builders = [ha.sobel_filter(im, slots=o.slots, init_level=6) for im in imgs]
    cst, hv, info = builders[0].assemble()  # the program does not depend on the data
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    inputs = []
    for s, im in enumerate(imgs):
        hevm.select_stream(s)
        hevm.setInput(0, im)
        inputs.append(_get_ct(hevm, ll, 0))
    assert not (inputs[0].data == inputs[1].data).all()
    hevm.run()
    assert hevm.stats()["keyswitches"] == 3 * (12 + 4)  # 9 rotations = 12 hops (offset 0 is free) + 4 ct*ct per stream
    for s in (2, 0, 1):
        hevm.select_stream(s)
        ovm.ciphers = [None] * len(ovm.ciphers)
        ovm.ciphers[0] = inputs[s]
        ovm.run()
        got = _get_ct(hevm, ll, ovm.prog.res_dst[0])
        want = ovm.ciphers[ovm.prog.res_dst[0]]
        assert got.ell == want.ell and (got.data == want.data).all()
        res = hevm.getOutput()[0]
        ref = builders[s].expected()[0]
        assert np.sqrt(np.mean((res - ref) ** 2)) < 1e-4 * max(1.0, np.abs(ref).max())


def test_reference_parameters_full_size_properties():
    """N = 2^15, 14 x 60-bit primes (SEAL_HEVM.cpp:39-53), default key set: size-independent properties of a program that
    touches every opcode -- rotate(k) then rotate(-k) is the identity, (x*y)*1 == x*y after rescale, x + (-x) == 0,
    a 6-hop NAF rotation equals np.roll, opcode 10 preserves the plaintext."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D)  # reference defaults
    assert hevm.logN == 15 and hevm.K == 14 and hevm.slots == 1 << 14
    rng = np.random.default_rng(21)
    x, y = rng.uniform(-1, 1, hevm.slots), rng.uniform(-1, 1, hevm.slots)
    b = ha.Builder(slots=hevm.slots, init_level=13)
    vx, vy = b.input(x), b.input(y)
    r = b.rotate(b.rotate(vx, 1365), -1365)            # 1365 = 0b10101010101: 6 NAF hops each way
    b.output(r)
    b.output(b.rotate(vy, 8191))                        # largest left rotation, NAF 8192 - 1
    p = b.mul(vx, vy)
    p = b.mul_plain(p, [1.0])                           # scale 2^120 -> rescale
    b.output(p)
    b.output(b.add(vx, b.negate(vx)))
    z = b.bootstrap(b.modswitch(p, 9), 4)
    b.output(b.add_plain(z, [0.5]))
    cst, hv, info = b.assemble()
    hevm.load_mem(cst, hv)
    hevm.setInput(0, x)
    hevm.setInput(1, y)
    hevm.run()
    res = hevm.getOutput()
    exp = b.expected()
    for got, want in zip(res, exp):  # key-switch noise at 13 primes and scale 2^40 is ~3e-6 per hop
        assert np.abs(got - want).max() < 2e-4
    assert np.abs(res[1] - np.roll(y, -8191)).max() < 2e-4
    st = hevm.stats()
    assert st["op_counts"][10] == 1 and st["keyswitches"] == 12 + 2 + 1


def test_edge_cases_constants_markers_and_noops(tmp_path):
    """ragged / scalar / over-long constants (src[i % len], SEAL_HEVM.cpp:259-261), the all-ones upscale constant (lhs 0xFFFF),
    buffer-allocation markers (opcode 0xFFFF, EmitHEVM.cpp:54-58), rotate by 0, modswitch by 0, unknown opcodes."""
    import ctypes

    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=7, logN=12, num_primes=4)
    o = Oracle(12, 4)
    slots = o.slots
    rng = np.random.default_rng(3)
    consts = [np.array([0.75]), rng.uniform(-1, 1, 3), rng.uniform(-1, 1, slots), rng.uniform(-1, 1, 4 * slots)]
    E, ROT, MSW, ADDCP, MULCP = ha.OP_ENCODE, ha.OP_ROTATE, ha.OP_MODSWITCH, ha.OP_ADDCP, ha.OP_MULCP
    ops = [(0xFFFF, 0, 0, 0)]                                 # tensor.empty marker
    for i in range(4):
        ops.append((E, i, i, (3 << 10) + 30))                 # plain regs 0..3 at level 3, scale 2^30
    ops += [(E, 4, 0xFFFF, (3 << 10) + 20),                   # all-ones constant at scale 2^20
            (ROT, 1, 0, 0),                                   # rotate by 0 into another register: a copy
            (MSW, 1, 1, 0),                                   # modswitch by 0: untouched
            (77, 1, 0, 0),                                    # unknown opcode: no-op
            (MULCP, 1, 1, 4),                                 # upscale by 2^20
            (ADDCP, 1, 1, 1)]                                 # + ragged constant... at the wrong scale? addcp forces lhs.scale = plain.scale
    hv = ha.pack_hevm([30], [3], [30], [3], [1], 2, 5, 3, np.array(ops, dtype=np.uint16))
    hevm.load_mem(ha.pack_cst(consts), hv)
    for i, cvec in enumerate(consts):                          # encoder tiling == oracle's, limb for limb (+-1 on a coefficient)
        lvl, sc = ctypes.c_int32(), ctypes.c_double()
        p = runner.lw.hevm_plain(hevm.vm, i, ctypes.byref(lvl), ctypes.byref(sc))
        got = ll.read_device(p, (lvl.value, o.N))
        want = o.encode(cvec, 2.0**30, 3)
        assert lvl.value == 3 and sc.value == 2.0**30
        d = o.ntt_inv(got, [0, 1, 2]).astype(np.int64) - o.ntt_inv(want.data, [0, 1, 2]).astype(np.int64)
        assert np.abs(d).max() <= 1
        assert np.abs(o.decode(Plaintext(got, 2.0**30)) - cvec[np.arange(slots) % len(cvec)]).max() < 1e-6
    x = rng.uniform(-1, 1, 5)                                  # ragged input, tiled the same way
    hevm.setInput(0, x)
    hevm.run()
    res = hevm.getOutput()[0]
    # register 1 = x * 1 (scale 2^50) then "+ c1" with the reference's scale overwrite: (x*2^50 + c1*2^30) / 2^30
    want = x[np.arange(slots) % 5] * 2.0**20 + consts[1][np.arange(slots) % 3]
    assert np.abs(res - want).max() < 1e-4 * 2.0**20  # input noise at scale 2^30, amplified by the 2^20 overwrite
    c = hevm.getCtxt(1)
    assert c.level == 3 and c.scale == 2.0**30


def test_large_batch_path_bit_exact(tmp_path):
    """64 independent rotations + ct*pt products of one ciphertext summed into one: one batched key-switch step large
    enough to take the throughput variants (separate inverse phase, base change + forward launch); limbs == oracle's."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=0x4845564D, logN=13, num_primes=5)
    o = Oracle(13, 5)
    _import_keys(o, hevm, ll)
    rng = np.random.default_rng(17)
    x = rng.uniform(-1, 1, o.slots)
    b = ha.Builder(slots=o.slots, init_level=4)
    v = b.modswitch(b.input(x), 1)  # level 3
    acc = None
    for k in range(64):
        r = b.rotate(v, [1, -1, 2, -2, 4, -4, 8, -8][k % 8] * (1 + k // 8 % 2 * 15))  # single- and two-hop offsets
        t = b.mul_plain(r, rng.uniform(-0.1, 0.1, 8), normalise=False)
        acc = t if acc is None else b.add(acc, t)
    sq = b.mul(acc, acc)
    b.output(sq)
    cst, hv, info = b.assemble()
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    hevm.setInput(0, x)
    ovm.ciphers[0] = _get_ct(hevm, ll, 0)
    hevm.run()
    ovm.run()
    got = _get_ct(hevm, ll, ovm.prog.res_dst[0])
    want = ovm.ciphers[ovm.prog.res_dst[0]]
    assert got.ell == want.ell and got.scale == want.scale and (got.data == want.data).all()
    res = hevm.getOutput()[0]
    assert np.sqrt(np.mean((res - b.expected()[0]) ** 2)) < 1e-4


def test_device_encoder_is_bit_identical_to_host_encoder():
    """preprocess() encodes every plaintext register in batched device kernels (encoder.hip) and decrypt() decodes on the
    device; option host_encoder = 1 keeps the one-at-a-time host FFTs.  Encoder: same algorithm, same operation order,
    no FMA contraction: identical limbs.  Decoder: same values to 1e-10."""
    import ctypes
    import os

    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    rng = np.random.default_rng(9)
    b = ha.Builder(slots=1 << 12, init_level=5, shadow=False)
    x = b.input(None)
    acc = None
    for n, (ln, sb, amp) in enumerate([(1, 40, 1.0), (7, 40, 3.0), (4096, 60, 1.0), (4096, 30, 1e-3), (5000, 40, 100.0), (16384, 50, 1.0),
                                       (3, 20, 7.5), (4096, 59, 0.5)]):
        t = b.mul_plain(x, rng.uniform(-amp, amp, ln), scale_bits=sb, normalise=False)
        t = b.add_plain(t, rng.uniform(-amp, amp, max(1, ln // 2)))
        acc = t if acc is None else acc
    b.output(b.upscale(acc, 20))  # the all-ones constant (lhs 0xFFFF)
    cst, hv, info = b.assemble()
    assert info["num_ptxt"] == 17
    plains, decoded = [], []
    for host in (False, True):
        hevm = runner.HEVM(seed=5, logN=13, num_primes=7, vm_options={"host_encoder": int(host)})
        hevm.load_mem(cst, hv)
        got = []
        for i in range(info["num_ptxt"]):
            lvl, sc = ctypes.c_int32(), ctypes.c_double()
            p = runner.lw.hevm_plain(hevm.vm, i, ctypes.byref(lvl), ctypes.byref(sc))
            assert p and lvl.value == 5
            got.append((ll.read_device(p, (lvl.value, hevm.N)), sc.value))
        plains.append(got)
        # same seed, same keys, same encryption randomness: the two VMs hold the same ciphertexts, so the outputs differ
        # only by the decoder (device: Garner digits -> double -> FFT kernels; host: SEAL's multi-precision compose + FFT)
        hevm.setInput(0, np.linspace(-1, 1, 4096))
        hevm.run()
        decoded.append(hevm.getOutput()[0])
    for (a, sa), (h, sh) in zip(*plains):
        assert sa == sh and (a == h).all()
    assert np.abs(decoded[0]).max() > 1e-3 and np.abs(decoded[0] - decoded[1]).max() < 1e-10 * max(1.0, np.abs(decoded[1]).max())


def test_rescale_operand_expressions_bit_exact(vm13, tmp_path):
    """Elementwise producers folded into the rescale that consumes them (plan mode; the other modes execute the same
    program op by op): a 40-term sum of ct*pt products and plain terms (more than one 16-product accumulation round),
    + plaintext, * plaintext, each combination feeding a rescale; ct*ct -> upscale -> rescale; everything bit-exact."""
    from dacapo_amd import hevm_asm as ha

    hevm, o, ll = vm13
    rng = np.random.default_rng(23)
    xv, yv = rng.uniform(-1, 1, o.slots), rng.uniform(-1, 1, o.slots)
    b = ha.Builder(slots=o.slots, init_level=6, shadow=True)
    x, y = b.input(xv), b.input(yv)
    w = lambda: rng.uniform(-0.2, 0.2, o.slots)  # noqa: E731

    def big_sum():  # single-use, so the whole sum is evaluated inside the rescale's loaders
        acc = None
        for k in range(40):
            src = x if k % 3 else y
            t = b.mul_plain(src, w(), scale_bits=60, normalise=False) if k % 5 else b.upscale(src, 60)
            acc = t if acc is None else b.add(acc, t)
        return acc

    r1 = b.rescale(big_sum())                                              # rescale(sum)
    r2 = b.rescale(b.add_plain(big_sum(), w()))                            # rescale(sum + pt)
    r3 = b.rescale(b.mul_plain(b.add_plain(r1, w()), w(), scale_bits=60, normalise=False))  # rescale((ct + pt) * pt)
    m = b.mul(r1, r2)                                                      # scale 80, level 5
    r4 = b.rescale(b.upscale(m, 20))                                       # the EVA upscale-then-rescale of a ct*ct product
    r5 = b.rescale(b.mul_plain(b.negate(r2), w(), scale_bits=60, normalise=False))
    out = b.add(b.add(b.modswitch(r3, 1), r4), b.modswitch(r5, 1))
    b.output(out)
    b.output(r1)  # r1 stays observable although it also feeds folded expressions
    cst, hv, info = b.assemble()
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    hevm.setInput(0, xv)
    hevm.setInput(1, yv)
    ovm.ciphers[0], ovm.ciphers[1] = _get_ct(hevm, ll, 0), _get_ct(hevm, ll, 1)
    hevm.run()
    ovm.run()
    for r in ovm.prog.res_dst:
        got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
        assert got.ell == want.ell and got.scale == want.scale and (got.data == want.data).all()
    res = hevm.getOutput()
    for k, ref in enumerate(b.expected()):
        assert np.sqrt(np.mean((res[k] - ref) ** 2)) < 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_random_programs_bit_exact(vm13, tmp_path, seed):
    """Differential test on random dataflow graphs: every deterministic opcode, shared and dead values, register recycling by
    the assembler, outputs that are also operands, chains the plan folds (sums, ct*pt into sums, producers into rescales),
    multi-hop rotations, modswitch views -- final ciphertext limbs of every result against the oracle VM."""
    from dacapo_amd import hevm_asm as ha

    hevm, o, ll = vm13
    rng = np.random.default_rng(1000 + seed)
    b = ha.Builder(slots=o.slots, init_level=6, shadow=True)
    ins = [b.input(rng.uniform(-1, 1, o.slots)) for _ in range(2)]
    pool = list(ins)
    vec = lambda: rng.uniform(-0.5, 0.5, int(rng.choice([1, 3, o.slots])))  # noqa: E731

    def pick(max_scale=None, min_level=1):
        cand = [v for v in pool if v.level >= min_level and (max_scale is None or v.scale_bits <= max_scale)]
        return cand[int(rng.integers(len(cand)))] if cand else None

    for _ in range(70):
        kind = rng.choice(["rot", "neg", "addcc", "addcp", "mulcp", "mulcc", "rescale", "modswitch", "rescale"])
        v = None
        if kind == "rot":
            x = pick()
            v = b.rotate(x, int(rng.choice([1, -1, 3, 5, -7, 12, 33, -100, 255])))
        elif kind == "neg":
            v = b.negate(pick())
        elif kind == "addcp":
            x = pick(max_scale=100)  # the plaintext is encoded at the ciphertext's scale; coefficients must fit 120 bits
            if x is not None:
                v = b.add_plain(x, vec())
        elif kind == "mulcp":
            x = pick(max_scale=100, min_level=2)
            if x is not None and x.scale_bits + 60 + 8 <= 60 * x.level:
                v = b.mul_plain(x, vec(), scale_bits=60, normalise=False)
        elif kind == "mulcc":
            x, y = pick(max_scale=60, min_level=2), pick(max_scale=60, min_level=2)
            if x is not None and y is not None:
                lv = min(x.level, y.level)
                if x.scale_bits + y.scale_bits + 8 <= 60 * lv:
                    x, y = b.modswitch(x, x.level - lv), b.modswitch(y, y.level - lv)
                    v = b._new(lv, x.scale_bits + y.scale_bits, x.plain * y.plain)
                    b._emit(ha.OP_MULCC, v, x, y.id, True)
        elif kind == "rescale":
            cand = [u for u in pool if u.scale_bits >= 100 and u.level >= 2]
            if cand:
                v = b.rescale(cand[int(rng.integers(len(cand)))])
        elif kind == "modswitch":
            x = pick(min_level=3)
            if x is not None:
                v = b.modswitch(x, 1)
        elif kind == "addcc":
            x = pick()
            same = [u for u in pool if u.level == x.level and u.scale_bits == x.scale_bits and u is not x]
            if same:
                y = same[int(rng.integers(len(same)))]
                v = b._new(x.level, x.scale_bits, x.plain + y.plain)
                b._emit(ha.OP_ADDCC, v, x, y.id, True)
        if v is not None and float(np.abs(v.plain).max()) < 1e3:
            pool.append(v)
    outs = [pool[int(i)] for i in rng.choice(len(pool), size=min(4, len(pool)), replace=False)]
    for u in outs:
        b.output(u)
    cst, hv, info = b.assemble(preserve_args=True)
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    for i, x in enumerate(ins):
        hevm.setInput(i, x.plain)
        ovm.ciphers[i] = _get_ct(hevm, ll, i)
    hevm.run()
    ovm.run()
    for r in ovm.prog.res_dst:
        got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
        assert got.ell == want.ell and got.scale == want.scale and (got.data == want.data).all(), (seed, r, info["op_mix"])


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_lazy_programs_with_bootstraps(vm13, seed):
    """Random programs lowered with the lazy policy at 3-prime bootstraps (what the traced ResNet-20 uses): many opcode 10,
    batched per wave in plan mode, one at a time otherwise; decrypted results against the cleartext shadow."""
    from dacapo_amd import hevm_asm as ha

    hevm, o, ll = vm13
    rng = np.random.default_rng(2000 + seed)
    b = ha.Builder(slots=o.slots, init_level=3, policy="lazy", boot_level=3, shadow=True)
    ins = [b.input(rng.uniform(-1, 1, o.slots)) for _ in range(2)]
    pool = list(ins)
    for _ in range(60):
        kind = rng.choice(["rot", "mulcp", "mulcc", "mulcc", "add", "addcp", "neg"])
        x = pool[int(rng.integers(len(pool)))]
        if kind == "rot":
            v = b.rotate(x, int(rng.choice([1, -3, 17, 64, -255])))
        elif kind == "mulcp":
            v = b.mul_plain(x, rng.uniform(-1, 1, o.slots))
        elif kind == "mulcc":
            v = b.mul(x, pool[int(rng.integers(len(pool)))])
        elif kind == "add":
            v = b.add(x, pool[int(rng.integers(len(pool)))])
        elif kind == "addcp":
            v = b.add_plain(x, rng.uniform(-1, 1, 5))
        else:
            v = b.negate(x)
        if float(np.abs(v.plain).max()) < 50:
            pool.append(v)
    outs = pool[-3:]
    for u in outs:
        b.output(b.finish(u))
    cst, hv, info = b.assemble()
    assert info["op_mix"]["bootstrap"] >= 1
    hevm.load_mem(cst, hv)
    for i, x in enumerate(ins):
        hevm.setInput(i, x.plain)
    hevm.run()
    res = hevm.getOutput()
    for k, ref in enumerate(b.expected()):
        assert np.sqrt(np.mean((res[k] - ref) ** 2)) < 2e-5 * max(1.0, float(np.abs(ref).max())), (seed, k, info["op_mix"])


@pytest.mark.parametrize("plan", [1, 0])
def test_zero_hop_rotate_is_a_copy_not_an_alias(plan):
    """rotate by 0 into ANOTHER register copies (rotate_vector, SEAL_HEVM.cpp:273).  The reference's addcp then overwrites the
    scale of its lhs register only (:308): the copy must keep the old scale, in the plan (SSA values) as in the eager loop."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=7, logN=12, num_primes=4, vm_options={"plan": plan})
    E, ROT, ADDCP = ha.OP_ENCODE, ha.OP_ROTATE, ha.OP_ADDCP
    ops = [(E, 0, 0, (3 << 10) + 20),   # plaintext at scale 2^20
           (ROT, 1, 0, 0),              # r1 = copy of r0 (scale 2^30)
           (ADDCP, 2, 0, 0)]            # r2 = r0 + pt: r0.scale := 2^20 (r1 must not follow)
    hv = ha.pack_hevm([30], [3], [30, 20], [3, 3], [1, 2], 3, 1, 3, np.array(ops, dtype=np.uint16))
    hevm.load_mem(ha.pack_cst([np.array([0.5])]), hv)
    x = np.linspace(-1, 1, hevm.slots)
    hevm.setInput(0, x)
    hevm.run()
    assert hevm.getCtxt(1).scale == 2.0**30 and hevm.getCtxt(2).scale == 2.0**20
    res = hevm.getOutput()
    assert np.abs(res[0] - x).max() < 1e-5


@pytest.mark.parametrize("case", ["cst_count", "cst_veclen", "hevm_nops", "plain_reg", "res_dst"])
def test_malformed_program_files_abort_with_a_message(tmp_path, case):
    """hostile / truncated .cst and .hevm images stop at load with a diagnostic (the reference reads them unchecked,
    SEAL_HEVM.cpp:182-234): negative or overflowing counts, an operation count the file cannot hold, operands naming
    registers that do not exist."""
    import struct
    import subprocess
    import sys

    from dacapo_amd import hevm_asm as ha

    cst = ha.pack_cst([np.array([0.5])])
    ops = [(ha.OP_ENCODE, 0, 0, (3 << 10) + 20), (ha.OP_ADDCP, 1, 0, 0)]
    hv = ha.pack_hevm([30], [3], [20], [3], [1], 2, 1, 3, np.array(ops, dtype=np.uint16))
    if case == "cst_count":
        cst = struct.pack("<q", -1) + cst[8:]
    elif case == "cst_veclen":
        cst = cst[:8] + struct.pack("<q", (1 << 61) + 1) + cst[16:]
    elif case == "hevm_nops":
        hv = hv[:32] + struct.pack("<Q", 1 << 40) + hv[40:]
    elif case == "plain_reg":
        bad = [(ha.OP_ENCODE, 0, 0, (3 << 10) + 20), (ha.OP_MULCP, 1, 0, 7)]
        hv = ha.pack_hevm([30], [3], [20], [3], [1], 2, 1, 3, np.array(bad, dtype=np.uint16))
    elif case == "res_dst":
        hv = ha.pack_hevm([30], [3], [20], [3], [1 << 20], 2, 1, 3, np.array(ops, dtype=np.uint16))
    (tmp_path / "p.cst").write_bytes(cst)
    (tmp_path / "p.hevm").write_bytes(hv)
    code = f"""
import sys
sys.path.insert(0, {str(__import__('pathlib').Path(__file__).resolve().parent.parent)!r})
from dacapo_amd import runner
h = runner.HEVM(seed=1, logN=12, num_primes=4)
h.load({str(tmp_path / 'p.cst')!r}, {str(tmp_path / 'p.hevm')!r})
print("loaded", flush=True)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "loaded" not in r.stdout and "[dacapo_amd]" in r.stderr, (r.stdout, r.stderr[-500:])


def test_direct_rotation_keys_make_rotations_single_hops(tmp_path):
    """hevm_add_rotation_keys = KeyGenerator::create_galois_keys(steps): with a key for the offset itself, rotate_vector is one key
    switch (Evaluator::rotate_internal takes the direct key; the HEaaN runtime's key list, HEAAN_HEVM.cpp:58-64, exists for that).
    Limbs equal the oracle's with the same keys; without the extra keys the same program takes one hop per NAF digit."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    hevm = runner.HEVM(seed=11, logN=13, num_primes=5)
    o = Oracle(13, 5)
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, o.slots)
    b = ha.Builder(slots=o.slots, init_level=4)
    v = b.input(x)
    y = b.add(b.rotate(v, 37), b.rotate(v, -100))      # NAF: 3 + 3 hops under the default key set
    b.output(b.add(y, b.rotate(y, 1000)))              # 1000 = 1024 - 32 + 8: 3 hops
    cst, hv, _ = b.assemble()
    hevm.load_mem(cst, hv)
    hevm.setInput(0, x)
    hevm.run()
    assert hevm.stats()["keyswitches"] == 9
    base = hevm.getOutput()[0]
    hevm.addRotationKeys([37, -100, o.slots + 1000])   # offsets are taken modulo the slot count
    _import_keys(o, hevm, ll)
    for step in (37, -100, 1000):
        elt = o.elt_from_step(step)
        p = runner.lw.hevm_galois_key(hevm.vm, elt)
        assert p, f"no direct key for step {step}"
        o.galois[elt] = ll.read_device(p, (o.K - 1, 2, o.K, o.N))
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    hevm.setInput(0, x)
    ovm.ciphers[0] = _get_ct(hevm, ll, 0)
    hevm.run()
    ovm.run()
    assert hevm.stats()["keyswitches"] == 3
    r = ovm.prog.res_dst[0]
    got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
    assert got.ell == want.ell and (got.data == want.data).all()
    res = hevm.getOutput()[0]
    assert np.sqrt(np.mean((res - b.expected()[0]) ** 2)) < 1e-5 and np.abs(res - base).max() < 1e-4


def test_online_encode_is_bit_identical_to_the_pre_encoded_pool():
    """option online_encode = 1 (the HEaaN runtime's way, HEAAN_HEVM.cpp:266-281,353-363): constants stay resident as doubles
    and every plaintext register is encoded right before the wave that first reads it, into a window recycled wave by wave.  Same
    seed, same program, same input: the result limbs equal the pre-encoded run's bit for bit, and the plaintext footprint shrinks."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    rng = np.random.default_rng(31)
    slots = 1 << 12
    b = ha.Builder(slots=slots, init_level=6, policy="lazy", boot_level=3, shadow=True)
    x = b.input(rng.uniform(-1, 1, slots))
    acc = None
    for k in range(24):                                   # a convolution-like layer: rotations x plaintext masks, summed, then squared
        t = b.mul_plain(b.rotate(x, [1, -1, 2, -2, 8, -8][k % 6] * (1 + k // 6)), rng.uniform(-0.2, 0.2, slots))
        acc = t if acc is None else b.add(acc, t)
    y = b.add_plain(b.mul(acc, acc), rng.uniform(-0.1, 0.1, slots))
    z = b.mul_plain(b.mul(y, x), [0.5])
    b.output(b.finish(b.add(z, b.mul_plain(y, rng.uniform(-1, 1, 7)))))
    cst, hv, info = b.assemble()
    res = {}
    for mode in ("0", "1"):
        hevm = runner.HEVM(seed=77, logN=13, num_primes=7, vm_options={"online_encode": int(mode)})
        hevm.load_mem(cst, hv)
        hevm.setInput(0, x.plain)
        hevm.run()
        hevm.run()                                        # graph replay re-encodes into the same window
        res[mode] = (_get_ct(hevm, ll, hevm.getResIdx(0)), hevm.plaintextBytes(), hevm.getOutput()[0])
    a, bb = res["0"][0], res["1"][0]
    assert a.ell == bb.ell and a.scale == bb.scale and (a.data == bb.data).all()
    assert res["1"][1] > 0 and res["0"][1] > 0   # (this program reads all its plaintexts in two waves: the footprint test is the ResNet one)
    assert np.sqrt(np.mean((res["1"][2] - b.expected()[0]) ** 2)) < 1e-4


def test_destroy_returns_the_vms_memory():
    """hevm_destroy (extension; the reference's VMs are never freed): after close() the device holds what it held before the VM was
    created, and a new VM of the same shape works -- five cycles do not accumulate."""
    import ctypes

    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    L = ll.lib()
    L.dc_mem_info.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.dc_mem_info.restype = None

    def free_bytes():
        L.dc_device_sync()
        f, t = ctypes.c_uint64(), ctypes.c_uint64()
        L.dc_mem_info(ctypes.byref(f), ctypes.byref(t))
        return f.value

    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, 1 << 13)
    b = ha.Builder(slots=1 << 13, init_level=4)
    v = b.input(x)
    y = b.mul(v, v)
    b.output(b.add(y, b.rotate(y, 5)))
    cst, hv, _ = b.assemble()

    def cycle(seed):
        hevm = runner.HEVM(seed=seed, logN=14, num_primes=6)
        hevm.addRotationKeys([5])
        hevm.load_mem(cst, hv)
        hevm.setInput(0, x)
        hevm.run()
        out = hevm.getOutput()[0]
        hevm.close()
        return out

    out = cycle(1)  # the first VM also pays for one-time runtime allocations (code objects, the library's own pools)
    assert np.sqrt(np.mean((out - b.expected()[0]) ** 2)) < 1e-5
    free0 = free_bytes()
    for seed in range(2, 6):
        out = cycle(seed)
        assert np.sqrt(np.mean((out - b.expected()[0]) ** 2)) < 1e-5
    free1 = free_bytes()
    one_vm = 6 * 2 * 6 * (1 << 14) * 8 * 5  # a lower bound on one VM's keys alone (~35 MB)
    assert free0 - free1 < one_vm, f"{(free0 - free1) / 1e6:.1f} MB still held after four create/destroy cycles"


@pytest.mark.parametrize("min_wgs", [0, 10**12])
def test_n_ary_sum_kernels_match_the_oracle_vm(tmp_path, min_wgs):
    """batch_ops.hip b_sum: a convolution-shaped sum (19 ciphertext x plaintext products + 9 bare ciphertexts: more than one 16-term and one
    8-term reduction window) in both forms of the kernel -- both polynomials of an item in one thread (large launches) and one polynomial
    per workgroup -- equals the oracle VM's multiply_plain / add sequence limb for limb.  The launch-shape option is flipped in-process
    (hevm_set_option; round 3 had to fork a child per setting because the threshold was latched from the environment)."""
    from dacapo_amd import runner
    from gpu_helpers import run_conv_shaped_program

    with runner.options(sum_pair_min_wgs=min_wgs):
        res = run_conv_shaped_program(13, 5, tmp_path)
    assert res["limbs_identical"] and res["scale_identical"] and res["max_error_vs_cleartext"] < 1e-4
    assert res["op_mix"]["mulcp"] >= 20 and res["op_mix"]["addcc"] >= 27


@pytest.mark.parametrize("min_wgs", [0, -1])
def test_sums_sharing_sources_match_the_oracle_vm(tmp_path, capfd, min_wgs):
    """batch_ops.hip b_sum_group_kernel (up to 8 sums of a step to a thread, the union of their sources read once) against one launch item per
    sum (option sum_group_min_wgs = -1): 11 output channels over 12 shared rotated inputs -- channels that skip taps, one that names a tap twice,
    bare ciphertext terms, a group of 8 and one of 3 -- both equal the oracle VM's multiply_plain / add sequence limb for limb; the plan's trace
    says which form ran."""
    import re

    from dacapo_amd import runner
    from gpu_helpers import run_multi_output_conv_program

    with runner.options(sum_group_min_wgs=min_wgs, trace=1):
        res = run_multi_output_conv_program(13, 5, tmp_path)
    err = capfd.readouterr().err
    m = re.search(r"sums sharing sources: (\d+) steps run (\d+) items as (\d+) groups", err)
    assert m, err[-2000:]
    if min_wgs == 0:
        assert int(m.group(1)) >= 1 and int(m.group(2)) >= 11 and int(m.group(3)) < int(m.group(2))
    else:
        assert int(m.group(1)) == 0
    assert res["limbs_identical"] and res["scale_identical"] and res["max_error_vs_cleartext"] < 1e-4
    assert res["op_mix"]["mulcp"] >= 100


@pytest.mark.parametrize("min_wgs", [0, 10**12])
def test_merged_and_fine_launch_shapes_of_the_key_switch_match_the_oracle_vm(tmp_path, min_wgs):
    """fused_ks.hip: the MERGE instantiations of L2 / L6 (one inverse COLS phase per source limb for all its target moduli) and of the fused
    L3-L5 kernel (both special-prime accumulators in one workgroup row) against the fine-grained ones, each forced for every launch of a
    program with 29 rotations (multi-hop), a ct x ct product and rescales at several levels: both equal the oracle VM limb for limb."""
    from dacapo_amd import runner
    from gpu_helpers import run_conv_shaped_program

    with runner.options(ks_merge_lift_min_wgs=min_wgs, ks_merge_special_min_wgs=min_wgs):
        for logN, K in ((13, 5), (12, 7)):
            res = run_conv_shaped_program(logN, K, tmp_path)
            assert res["limbs_identical"] and res["scale_identical"] and res["max_error_vs_cleartext"] < 1e-4
            assert res["stats"]["keyswitches"] >= 40


@pytest.mark.parametrize("opts", [dict(ks_items_fast=0), dict(cols_pairs=0), dict(tiny_tile_wgs=512), dict(tiny_tile_wgs=100000),
                                  dict(ks_items_fast=0, cols_pairs=0, tiny_tile_wgs=512, ks_big_tiles=0), dict(ks_big_tiles=0, ks_merge_lift_min_wgs=0)])
def test_round5_launch_shapes_are_second_implementations_of_the_same_limbs(tmp_path, opts):
    """round 5's launch-shape options, each switched off (or forced) for every launch of the convolution-shaped program -- 29 rotations whose
    NAF hops name 7 Galois elements several times each (what ks_items_fast sorts by), a ct x ct product, rescales at several levels: the
    key-ordered items and the (tiles, items, rows) grid, twiddle pairs in the forward COLS tiles, the one-butterfly tile geometry for every
    launch / for none, the large-batch sequence (separate inverse COLS + lift launches) forced at every size.  All equal the oracle VM limb for limb."""
    from dacapo_amd import runner
    from gpu_helpers import run_conv_shaped_program

    with runner.options(**opts):
        for logN, K in ((13, 5), (12, 7)):
            res = run_conv_shaped_program(logN, K, tmp_path)
            assert res["limbs_identical"] and res["scale_identical"] and res["max_error_vs_cleartext"] < 1e-4, (opts, logN, K)


def test_options_table_round_trips_and_rejects_nothing_silently():
    """hevm_set_option / hevm_get_option / hevm_reset_options (include/hevm_abi.h): a value set is the value read, a with-block restores,
    reset returns to the documented defaults -- the reference's ring among them (SEAL_HEVM.cpp:39-40: N = 2^15, 14 primes)"""
    from dacapo_amd import runner

    assert runner.get_option("logn") == 15 and runner.get_option("primes") == 14 and runner.get_option("plan") == 1
    with runner.options(sum_pair_min_wgs=7, plan=0):
        assert runner.get_option("sum_pair_min_wgs") == 7 and runner.get_option("plan") == 0
    assert runner.get_option("sum_pair_min_wgs") == 1024 and runner.get_option("plan") == 1
    runner.set_option("max_batch", 16)
    runner.lw.hevm_reset_options()
    assert runner.get_option("max_batch") == 64


@pytest.mark.parametrize("plan,ks", [(1, 1), (0, 1), (1, 2)])
def test_bounded_rotation_key_set_serves_every_offset(tmp_path, plan, ks):
    """option rot_compose (extension; the reference's HEaaN runtime serves every rotation of a program from 49 left-rotation keys,
    HEAAN_HEVM.cpp:58-64,124-126): a rotation without a direct key is the SHORTEST sum of offsets that have one, found in a fixed order
    (HEVM::compose_rotation; restated in oracle/oracle.py).  Key set = SEAL's default +-2^k plus a few of that list's other offsets; the
    program rotates by offsets outside it.  GPU VM == oracle VM limb for limb (plan and one-at-a-time loop, SEAL-style and grouped-digit
    keys), fewer key switches than SEAL's NAF over the power-of-two keys would take, and the slots land where they should."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    logN, K = 12, 5 + ks
    slots = 1 << (logN - 1)
    extra = [3, 5, 6, 7, 24, 96, 160, 192, 224, 768]
    offsets = [9, 11, 100, 777, -3, 1000, 23, -97, 455]
    rng = np.random.default_rng(2)
    b = ha.Builder(slots=slots, init_level=K - ks, policy="lazy", boot_level=K - ks, shadow=True)
    x = b.input(rng.uniform(-1, 1, slots))
    acc = None
    for k, o in enumerate(offsets):
        t = b.mul_plain(b.rotate(x, o), [0.1 * (k + 1)])
        acc = t if acc is None else b.add(acc, t)
    b.output(b.finish(acc))
    cst, hv, _ = b.assemble()
    hevm = runner.HEVM(seed=13, logN=logN, num_primes=K, ks_special=ks, vm_options={"plan": plan, "rot_compose": 1})
    hevm.addRotationKeys(extra)
    o = Oracle(logN, K)
    if ks > 1:
        o.set_hybrid(ks)
    o.rot_compose = True
    elts = sorted(set(o.default_galois_elts()) | {o.elt_from_step(s) for s in extra})
    _import_keys(o, hevm, ll, elts=elts)
    hops = {off: o.rotate_hops(off) for off in offsets}
    assert all(1 <= len(h) <= 3 for h in hops.values()) and len(hops[9]) == 2 and len(hops[777]) == 3
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    hevm.setInput(0, x.plain)
    ovm.ciphers[0] = _get_ct(hevm, ll, 0)
    hevm.run()
    ovm.run()
    r = ovm.prog.res_dst[0]
    got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
    assert got.ell == want.ell and got.scale == want.scale and (got.data == want.data).all()
    assert np.abs(hevm.getOutput()[0] - b.expected()[0]).max() < 1e-4
    # the plan computes a hop (value, Galois element) once: composed rotations of one value that start with the same part share it
    # (plan_exec.hip hop_memo; the same limbs, a key switch being deterministic); the one-at-a-time loop executes every hop
    prefixes = {tuple(h[: k + 1]) for h in hops.values() for k in range(len(h))}
    total = sum(len(h) for h in hops.values())
    assert len(prefixes) < total, "the offsets of this test are meant to share first hops"
    assert hevm.stats()["keyswitches"] == (len(prefixes) if plan else total)
    o.rot_compose = False
    assert sum(len(o.rotate_hops(off)) for off in offsets) > total                              # SEAL's NAF over +-2^k takes more hops
    hevm.close()


@pytest.mark.parametrize("plan", [1, 0])
def test_duplicate_rotations_keep_their_own_scale_metadata_under_rot_compose(tmp_path, plan):
    """option rot_compose memoises a hop (value, Galois element): two rotations of one value by one offset name the same limbs.  The
    reference's scale overwrites (addcc sets lhs.scale = rhs.scale on ONE register, SEAL_HEVM.cpp:301) must not reach the other register,
    and a rotation issued after such an overwrite carries its operand's scale, not the memoised value's (round-4 advisor finding on
    plan_exec.hip hop_memo).  Raw program: r2 = rot(x, 9); r3 = rot(x, 9); r4 = r2 + y (y at another scale: r2's label is overwritten);
    r5 = rot(x, 9); r6 = r3 + y.  Every result register: level, scale label and limbs == the oracle VM with the same composition rule."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    logN, K = 12, 5
    ROT, ADDCC = ha.OP_ROTATE, ha.OP_ADDCC
    ops = [(ROT, 2, 0, 9), (ROT, 3, 0, 9), (ADDCC, 4, 2, 1), (ROT, 5, 0, 9), (ADDCC, 6, 3, 1), (ROT, 7, 2, 9)]
    res = [2, 3, 4, 5, 6, 7]
    hv = ha.pack_hevm([30, 25], [4, 4], [30] * len(res), [4] * len(res), res, 8, 1, 4, np.array(ops, dtype=np.uint16))
    cst = ha.pack_cst([np.array([1.0])])
    hevm = runner.HEVM(seed=17, logN=logN, num_primes=K, vm_options={"plan": plan, "rot_compose": 1})
    hevm.addRotationKeys([3, 6])  # 9 = 3 + 6 under the bounded set (two hops, the first shared by every rotation here)
    o = Oracle(logN, K)
    o.rot_compose = True
    elts = sorted(set(o.default_galois_elts()) | {o.elt_from_step(s) for s in (3, 6)})
    _import_keys(o, hevm, ll, elts=elts)
    assert len(o.rotate_hops(9)) == 2
    hevm.load_mem(cst, hv)
    ovm = _mirror_vm(hevm, ll, o, cst, hv, tmp_path)
    rng = np.random.default_rng(4)
    for i in range(2):
        hevm.setInput(i, rng.uniform(-1, 1, 1 << (logN - 1)))
        ovm.ciphers[i] = _get_ct(hevm, ll, i)
    hevm.run()
    ovm.run()
    for r in res:
        got, want = _get_ct(hevm, ll, r), ovm.ciphers[r]
        assert got.ell == want.ell and got.scale == want.scale, (r, got.scale, want.scale)
        assert (got.data == want.data).all(), r
    assert _get_ct(hevm, ll, 2).scale == 2.0**25 and _get_ct(hevm, ll, 3).scale == 2.0**25  # both overwritten, each by its own addcc
    assert _get_ct(hevm, ll, 5).scale == 2.0**30 and _get_ct(hevm, ll, 7).scale == 2.0**25  # operand's scale at the time of the rotation
    if plan:  # x -> 9 costs two hops once; r7 rotates r2 (another value): two more
        assert hevm.stats()["keyswitches"] == 4
    hevm.close()


def test_explicit_dag_equals_captured_graph_with_fused_links_and_opcode10(tmp_path):
    """plan_graph = 2 (the plan's graph built node by node from its own dependencies) against plan_graph = 1 (captured from two streams) on
    a program with chain-fusion links (ct x ct -> rescale, rescale -> opcode 10) and opcode 10 in it, zero-encryption hook on so that the
    results are deterministic: identical limbs in every result register, run twice (the epoch bump sits at the graph's tail in both forms,
    where issue_plan puts it: round-4 advisor finding on capture_plan_dag)."""
    from dacapo_amd import hevm_asm as ha
    from dacapo_amd import lowlevel as ll
    from dacapo_amd import runner

    logN, K = 12, 5
    slots = 1 << (logN - 1)
    rng = np.random.default_rng(8)
    b = ha.Builder(slots=slots, init_level=K - 1, policy="lazy", boot_level=2, shadow=True)
    x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
    p = b.mul(x, y)
    q = b.mul(b.rotate(p, 3), b.add(p, x))
    r = b.mul(q, q)
    r = b.mul(r, b.rotate(r, 5))   # runs out of primes on the way: the lazy policy re-encrypts (opcode 10)
    b.output(b.finish(b.add(r, b.mul_plain(x, [0.5]))))
    cst, hv, info = b.assemble()
    assert info["op_mix"].get("bootstrap", 0) >= 1
    outs = {}
    for mode in (1, 2):
        hevm = runner.HEVM(seed=23, logN=logN, num_primes=K, vm_options={"plan_graph": mode})
        runner.lw.hevm_test_zero_encryption(hevm.vm, True)
        hevm.load_mem(cst, hv)
        for i, a in enumerate(b.args):
            hevm.setInput(i, a.plain)
        runs = []
        for _ in range(2):
            hevm.run()
            runs.append(_get_ct(hevm, ll, hevm.res_idx(0) if hasattr(hevm, "res_idx") else int(hevm.lw.getResIdx(hevm.vm, 0))))
        assert runs[0].ell == runs[1].ell and (runs[0].data == runs[1].data).all()
        st = hevm.stats()
        outs[mode] = (runs[1], st["keyswitches"])
        hevm.close()
    assert outs[1][1] == outs[2][1]
    assert outs[1][0].ell == outs[2][0].ell and outs[1][0].scale == outs[2][0].scale and (outs[1][0].data == outs[2][0].data).all()
