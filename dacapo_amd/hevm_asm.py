"""HEVM program assembler: writes the reference's `.hevm` bytecode and `.cst` constant files.

The reference produces these files with an MLIR pass pipeline (`hecate-opt ... --emit-hevm`,
/root/reference/lib/Dialect/CKKS/Transforms/EmitHEVM.cpp:28-120 and
lib/Dialect/Earth/Transforms/ElideConstant.cpp:40-53) that cannot be built here (no MLIR).  This module writes
the same wire format (include/hecate/Support/HEVMHeader.h:10-35) from a small tracing builder so that the
runtime can be exercised with real programs:

  * operand encoding follows include/hecate/Dialect/CKKS/IR/CKKSOps.td:69-222 (opcode numbers, rhs packing
    `(level << 10) + scale` for encode, int16 rotation offsets, 0xFFFF = all-ones "upscale" constant);
  * scale management is a minimal EVA-style waterline policy (rescale once the scale would stay >= the waterline,
    modswitch to equalise levels, upscale = mulcp by the all-ones constant: lib/Conversion/.../UpscaleToMulcp.cpp:52-72);
  * cipher registers are re-used once dead, arguments first, like ReuseBuffer.cpp:27-55 / EmitHEVM.cpp:45-52.

Every value also carries its plaintext slot vector (numpy), so the expected result of a program is known when it
is built (the examples/tests/*.py scripts recompute it the same way).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

MAGIC = 0x4845564D
OP_ENCODE, OP_ROTATE, OP_NEGATE, OP_RESCALE, OP_MODSWITCH, OP_UPSCALE = 0, 1, 2, 3, 4, 5
OP_ADDCC, OP_ADDCP, OP_MULCC, OP_MULCP, OP_BOOTSTRAP = 6, 7, 8, 9, 10
OP_NAMES = ["encode", "rotate", "negate", "rescale", "modswitch", "upscale", "addcc", "addcp", "mulcc", "mulcp", "bootstrap"]
# Extension opcodes of this runtime (not emitted by the reference's compiler; its VMs skip unknown opcodes, SEAL_HEVM.cpp:351-399).
# They exist so that real CKKS bootstrapping (dacapo_amd/ckks_boot.py; HEAAN_HEVM.cpp:386-399) is an instruction sequence like any other:
OP_ENCODE_COMPLEX = 16  # dst = plain reg, lhs = constant index (vector [re..., im...]), rhs = (level << 10) + scale: complex slot values
OP_CONJ = 17            # dst = complex conjugate of lhs (Galois element 2N - 1, a key of the default set)
OP_MODRAISE = 18        # dst = lhs (1 prime) re-read modulo the first rhs primes: decrypts to p + q0 * I
OP_SETSCALE = 19        # dst = lhs with its scale label set to constants[rhs][0]


def write_cst(path, constants):
    """ElideConstant.cpp:40-53: i64 count; {i64 len; f64 data[len]} * count."""
    with open(path, "wb") as f:
        f.write(struct.pack("<q", len(constants)))
        for c in constants:
            c = np.ascontiguousarray(c, dtype="<f8").ravel()
            f.write(struct.pack("<q", len(c)))
            f.write(c.tobytes())


def pack_cst(constants) -> bytes:
    out = [struct.pack("<q", len(constants))]
    for c in constants:
        c = np.ascontiguousarray(c, dtype="<f8").ravel()
        out += [struct.pack("<q", len(c)), c.tobytes()]
    return b"".join(out)


def pack_hevm(arg_scale, arg_level, res_scale, res_level, res_dst, num_ctxt, num_ptxt, init_level, ops) -> bytes:
    """EmitHEVM.cpp:31-37,94-119."""
    na, nr = len(arg_scale), len(res_scale)
    ops = np.ascontiguousarray(ops, dtype="<u2").reshape(-1, 4)
    body_len = 40 + 8 * (2 * na + 3 * nr)
    out = [struct.pack("<IIQQ", MAGIC, 24, na, nr), struct.pack("<5Q", body_len, len(ops), num_ctxt, num_ptxt, init_level)]
    for arr in (arg_scale, arg_level, res_scale, res_level, res_dst):
        out.append(struct.pack(f"<{len(arr)}Q", *[int(x) for x in arr]))
    out.append(ops.tobytes())
    return b"".join(out)


def unpack_hevm(raw: bytes) -> dict:
    """inverse of pack_hevm (HEVMHeader.h:10-35)"""
    magic, hsize, na, nr = struct.unpack_from("<IIQQ", raw, 0)
    assert magic == MAGIC and hsize == 24
    body_len, nops, nct, npt, init_level = struct.unpack_from("<5Q", raw, 24)
    off, arrs = 64, []
    for n in (na, na, nr, nr, nr):
        arrs.append(list(struct.unpack_from(f"<{n}Q", raw, off)))
        off += 8 * n
    ops = np.frombuffer(raw, dtype="<u2", count=4 * nops, offset=off).reshape(nops, 4).copy()
    return {"arg_scale": arrs[0], "arg_level": arrs[1], "res_scale": arrs[2], "res_level": arrs[3], "res_dst": arrs[4],
            "num_ctxt": nct, "num_ptxt": npt, "init_level": init_level, "ops": ops}


def truncate_hevm(raw: bytes, num_ops: int):
    """The first `num_ops` instructions of a program as a program of its own whose single result is the register
    the last kept ciphertext instruction wrote.  Returns (hevm_bytes, level, scale_bits) of that result, tracked
    through the instruction stream the way SEAL_HEVM.cpp:268-334 updates them."""
    h = unpack_hevm(raw)
    ops = h["ops"][:num_ops]
    lvl = {i: int(v) for i, v in enumerate(h["arg_level"])}
    scl = {i: int(v) for i, v in enumerate(h["arg_scale"])}
    plvl, pscl, last = {}, {}, None
    for opc, dst, lhs, rhs in ops.tolist():
        if opc == OP_ENCODE:
            plvl[dst], pscl[dst] = rhs >> 10, rhs & 0x3FF
            continue
        if opc > OP_BOOTSTRAP:
            continue
        l, sc = lvl[lhs], scl[lhs]
        if opc == OP_RESCALE:
            l, sc = l - 1, sc - 60
        elif opc == OP_MODSWITCH:
            l -= rhs
        elif opc == OP_ADDCC:
            sc = scl[rhs]
        elif opc == OP_ADDCP:
            sc = pscl[rhs]
        elif opc == OP_MULCC:
            sc += scl[rhs]
        elif opc == OP_MULCP:
            sc += pscl[rhs]
        elif opc == OP_BOOTSTRAP:
            l = rhs
        lvl[dst], scl[dst], last = l, sc, dst
    assert last is not None
    out = pack_hevm(h["arg_scale"], h["arg_level"], [scl[last]], [lvl[last]], [last], h["num_ctxt"], h["num_ptxt"],
                    h["init_level"], ops)
    return out, lvl[last], scl[last]


def unpack_cst(raw: bytes):
    (n,) = struct.unpack_from("<q", raw, 0)
    off, out = 8, []
    for _ in range(n):
        (ln,) = struct.unpack_from("<q", raw, off)
        off += 8
        out.append(np.frombuffer(raw, dtype="<f8", count=ln, offset=off))
        off += 8 * ln
    return out


def plain_eval(hevm: bytes, cst: bytes, inputs, slots=1 << 14):
    """What the program computes on cleartext slot vectors (no encryption, no noise): the expected value of a run,
    opcode semantics of SEAL_HEVM.cpp:268-334 with rescale / modswitch / opcode 10 as identities.  Returns the results."""
    h = unpack_hevm(hevm)
    consts = unpack_cst(cst)
    idx = np.arange(slots)
    tile = lambda v: np.asarray(v, dtype=np.float64).ravel()[idx % len(np.asarray(v).ravel())]  # noqa: E731
    reg = {i: tile(v) for i, v in enumerate(inputs)}
    plain = {}
    for opc, dst, lhs, rhs in h["ops"].tolist():
        if opc == OP_ENCODE:
            plain[dst] = np.ones(slots) if lhs == 0xFFFF else tile(consts[lhs])
        elif opc == OP_ROTATE:
            reg[dst] = np.roll(reg[lhs], -(rhs - 65536 if rhs >= 32768 else rhs))
        elif opc == OP_NEGATE:
            reg[dst] = -reg[lhs]
        elif opc in (OP_RESCALE, OP_BOOTSTRAP):
            reg[dst] = reg[lhs]
        elif opc == OP_MODSWITCH:
            if (rhs - 65536 if rhs >= 32768 else rhs) > 0:
                reg[dst] = reg[lhs]
        elif opc == OP_ADDCC:
            reg[dst] = reg[lhs] + reg[rhs]
        elif opc == OP_ADDCP:
            reg[dst] = reg[lhs] + plain[rhs]
        elif opc == OP_MULCC:
            reg[dst] = reg[lhs] * reg[rhs]
        elif opc == OP_MULCP:
            reg[dst] = reg[lhs] * plain[rhs]
    return [reg[d] for d in h["res_dst"]]


def read_fixture(prefix) -> dict:
    """A traced program committed as data (tests/golden/<name>.{hevm.gz,cst.xz,input.npz,json}; written by
    tools/fixtures/trace_reference_model.py): returns the decompressed `.hevm` / `.cst` bytes, the packed input and metadata."""
    import gzip
    import json
    import lzma
    prefix = str(prefix)
    fx = {"hevm": gzip.open(prefix + ".hevm.gz").read(), "cst": lzma.open(prefix + ".cst.xz").read(),
          "meta": json.loads(Path(prefix + ".json").read_text())}
    if Path(prefix + ".input.npz").exists():  # the ResNet fixture
        z = np.load(prefix + ".input.npz")
        fx.update(packed=z["packed"], torch_result=z["torch_result"], expected=z["expected"])
    else:  # benchmark-suite fixtures: inputs as the reference's test script fed them, expected = cleartext evaluation
        z = np.load(prefix + ".io.npz")
        fx.update(inputs=[z[f"input{i}"] for i in range(fx["meta"]["num_inputs"])], expected=z["expected"])
    return fx


@dataclass
class Value:
    """An SSA ciphertext value of the traced program."""
    id: int
    level: int
    scale_bits: int
    plain: np.ndarray | None  # expected slot values (None when plaintext shadowing is off)
    last_use: int = -1


@dataclass
class _Op:
    opcode: int
    dst: int | None  # value id (cipher) or plain register (encode)
    lhs: int
    rhs: int
    lhs_is_value: bool = True
    rhs_is_value: bool = False


class Builder:
    def __init__(self, slots=1 << 14, waterline=40, init_level=13, rescale_bits=60, min_level=1, shadow=True,
                 policy="eager", boot_level=None, headroom=16, rotate_reserve=0, carry_scale=False, real_boot=None):
        """policy "eager": rescale a product as soon as its scale allows (EVA's waterline rule; the caller places
        bootstraps).  policy "lazy": products keep their scale, sums of products are rescaled ONCE when the sum is next
        multiplied or rotated (what the reference's scale-management passes achieve by moving rescales below
        additions, EarthOps.td:350-564), and a value that runs out of primes is bootstrapped to `boot_level` primes.
        `headroom` = bits kept free above the scale for the magnitude of the slot values."""
        self.slots, self.waterline, self.init_level, self.rescale_bits = slots, waterline, init_level, rescale_bits
        self.min_level, self.shadow = min_level, shadow
        assert policy in ("eager", "lazy")
        self.lazy, self.headroom = policy == "lazy", headroom
        self.rotate_reserve = rotate_reserve  # lazy: scale bits a rotation's result must still be able to absorb
        # lazy, carry_scale: EVA's rule proper -- rescale only while the result stays at or above the waterline, and let a
        # value carry a scale between waterline and waterline + one prime into its next multiplication (a ct*ct product
        # at 2^80 is not upscaled to 2^100 just to be rescaled): ~2/3 of the rescales and bootstraps of the exact rule
        self.carry_scale = carry_scale
        self.boot_level = init_level if boot_level is None else boot_level
        # real_boot = dict(num_primes=K, r=5, msg_bits=7): `bootstrap` emits REAL CKKS bootstrapping (dacapo_amd/ckks_boot.py: ModRaise,
        # CoeffToSlot, EvalMod, SlotToCoeff over extension opcodes 16-19) instead of opcode 10, the SEAL VM's decrypt / re-encrypt stand-in
        self.real_boot, self._boot_emitter, self._scale_mirror = real_boot, None, None
        self._memo: dict = {}  # lazy policy: (kind, value id, arg) -> value, so a shared operand is rescaled/bootstrapped once
        self.values: list[Value] = []
        self.ops: list[_Op] = []
        self.constants: list[np.ndarray] = []
        self._const_index = {}
        self.num_plain = 0
        self.args: list[Value] = []
        self.results: list[Value] = []

    # ---- helpers ----------------------------------------------------------------------------------------------
    def _new(self, level, scale_bits, plain):
        if self.shadow and plain is not None:
            # CKKS capacity: |value| * scale must stay below Q_level / 2 or decryption wraps around
            peak = float(np.max(np.abs(plain))) if len(plain) else 0.0
            if peak > 0 and np.log2(peak) + scale_bits > self.rescale_bits * level - 2:
                raise OverflowError(f"value of magnitude {peak:.3g} at scale 2^{scale_bits} does not fit {level} prime(s)")
        v = Value(len(self.values), level, scale_bits, plain if self.shadow else None)
        self.values.append(v)
        return v

    def _emit(self, opcode, dst: Value, lhs: Value, rhs=0, rhs_is_value=False):
        idx = len(self.ops)
        lhs.last_use = max(lhs.last_use, idx)  # a value already declared an output stays live to the end
        if rhs_is_value:
            self.values[rhs].last_use = max(self.values[rhs].last_use, idx)
        self.ops.append(_Op(opcode, dst.id, lhs.id, rhs, True, rhs_is_value))

    def _tile(self, vec):
        vec = np.asarray(vec, dtype=np.float64).ravel()
        return vec[np.arange(self.slots) % len(vec)]

    def _const(self, vec) -> int:
        vec = np.ascontiguousarray(vec, dtype=np.float64).ravel()
        key = vec.tobytes()
        if key not in self._const_index:
            self._const_index[key] = len(self.constants)
            self.constants.append(vec)
        return self._const_index[key]

    def _encode(self, const_idx: int, level: int, scale_bits: int) -> int:
        """opcode 0: dst = plain register, lhs = constant index (0xFFFF = ones), rhs = (level << 10) + scale."""
        assert 0 < scale_bits < 1024 and 0 < level < 64
        reg = self.num_plain
        self.num_plain += 1
        self.ops.append(_Op(OP_ENCODE, reg, const_idx, (level << 10) + scale_bits, False, False))
        return reg

    # ---- program inputs / outputs ---------------------------------------------------------------------------------
    def input(self, plain=None, level=None, scale_bits=None) -> Value:
        assert not self.ops, "arguments occupy the first registers: declare them before any op"
        p = self._tile(plain) if (plain is not None and self.shadow) else None
        v = self._new(self.init_level if level is None else level, self.waterline if scale_bits is None else scale_bits, p)
        self.args.append(v)
        return v

    def output(self, v: Value):
        v.last_use = 1 << 60
        self.results.append(v)

    # ---- ciphertext ops ------------------------------------------------------------------------------------------
    def rotate(self, x: Value, offset: int) -> Value:
        assert -(1 << 15) <= offset < (1 << 15)
        if self.lazy:
            x = self._prepare(x, self.rotate_reserve)  # a key switch is cheapest at the lowest level the value can have
        out = self._new(x.level, x.scale_bits, np.roll(x.plain, -offset) if self.shadow else None)
        self._emit(OP_ROTATE, out, x, offset & 0xFFFF)
        return out

    def negate(self, x: Value) -> Value:
        out = self._new(x.level, x.scale_bits, -x.plain if self.shadow else None)
        self._emit(OP_NEGATE, out, x)
        return out

    def rescale(self, x: Value) -> Value:
        assert x.level > 1, "out of levels: bootstrap first"
        out = self._new(x.level - 1, x.scale_bits - self.rescale_bits, x.plain)
        self._emit(OP_RESCALE, out, x)
        return out

    def modswitch(self, x: Value, down: int) -> Value:
        if down <= 0:
            return x
        assert x.level - down >= 1
        if self.lazy and ("modswitch", x.id, down) in self._memo:
            return self._memo[("modswitch", x.id, down)]
        out = self._new(x.level - down, x.scale_bits, x.plain)
        self._emit(OP_MODSWITCH, out, x, down)
        if self.lazy:
            self._memo[("modswitch", x.id, down)] = out
        return out

    def upscale(self, x: Value, bits: int) -> Value:
        """UpscaleToMulcp: multiply by the all-ones plaintext encoded at scale 2^bits (constant index -1)."""
        if bits <= 0:
            return x
        reg = self._encode(0xFFFF, x.level, bits)
        out = self._new(x.level, x.scale_bits + bits, x.plain)
        self._emit(OP_MULCP, out, x, reg)
        return out

    def bootstrap(self, x: Value, target_level=None) -> Value:
        t = self.init_level if target_level is None else target_level
        if self.real_boot is not None:
            return self._real_bootstrap(x, t)
        if self.shadow and x.plain is not None:  # how large a bootstrapped value gets (sizes real_boot's msg_bits when opcode 10 is lowered later)
            self.boot_peaks = getattr(self, "boot_peaks", []) + [float(np.max(np.abs(x.plain)))]
        out = self._new(t, x.scale_bits, x.plain)
        self._emit(OP_BOOTSTRAP, out, x, t)
        return out

    def _real_bootstrap(self, x: Value, t: int) -> Value:
        from . import ckks_boot

        rb = self.real_boot
        logN = self.slots.bit_length()  # slots = N / 2
        if self._boot_emitter is None:
            self._boot_emitter = ckks_boot.BootstrapEmitter(self, logN, rb["num_primes"], t, r=rb.get("r", 5), msg_bits=rb.get("msg_bits", 0),
                                                            out_bits=self.waterline, ks=rb.get("ks", 1), primes=rb.get("primes"))
            self._scale_mirror = ckks_boot.ScaleMirror(self, self._boot_emitter.primes)
        em = self._boot_emitter
        assert t == em.target, "every real bootstrap of a program restores the same number of primes"
        assert x.scale_bits <= em.boot_in_bits, f"a value at scale 2^{x.scale_bits} cannot enter a bootstrap (limit 2^{em.boot_in_bits})"
        if self.shadow and x.plain is not None:
            peak = float(np.max(np.abs(x.plain)))
            assert peak < 2.0 ** rb.get("msg_bits", 0), f"message of magnitude {peak:.3g} exceeds real_boot['msg_bits']"
            rb["peak_seen"] = max(rb.get("peak_seen", 0.0), peak)
        scale = self._scale_mirror.upto()[x.id]
        v, _ = em.bootstrap(x, scale)
        v.scale_bits, v.plain = self.waterline, x.plain
        return v

    def _normalise(self, x: Value) -> Value:
        while x.scale_bits - self.rescale_bits >= self.waterline and x.level > self.min_level:
            x = self.rescale(x)
        return x

    # ---- lazy policy ---------------------------------------------------------------------------------------------
    def _fits(self, level: int, scale_bits: int) -> bool:
        return scale_bits + self.headroom <= self.rescale_bits * level

    def _memoised(self, kind, x: Value, arg, make) -> Value:
        key = (kind, x.id, arg)
        if key not in self._memo:
            self._memo[key] = make()
        return self._memo[key]

    def _boot(self, x: Value) -> Value:
        return self._memoised("boot", x, self.boot_level, lambda: self.bootstrap(x, self.boot_level))

    def _rescale_or_boot(self, x: Value) -> Value:
        if x.level <= self.min_level:
            x = self._boot(x)
        return self._memoised("rescale", x, 0, lambda: self.rescale(x))

    def _prepare(self, x: Value, extra_bits: int) -> Value:
        """bring x to the waterline exactly (whole primes by rescaling; a remainder by the EVA trick of upscaling to
        waterline + one prime first) and make sure an op that adds `extra_bits` of scale still fits its primes"""
        while x.scale_bits - self.rescale_bits >= self.waterline:
            x = self._rescale_or_boot(x)
        if x.scale_bits > self.waterline and not self.carry_scale:
            up = self.waterline + self.rescale_bits - x.scale_bits
            if x.level <= self.min_level or not self._fits(x.level, x.scale_bits + up):
                x = self._boot(x)
            y = x
            x = self._memoised("uprescale", y, up, lambda: self.rescale(self.upscale(y, up)))
        if not self._fits(x.level, x.scale_bits + extra_bits):
            x = self._boot(x)
            assert self._fits(x.level, x.scale_bits + extra_bits), "boot_level too small for this product"
        return x

    def hint(self, x: Value, need: int) -> Value:
        """a bootstrap HINT of the traced program (the reference's model scripts call hc.bootstrap before every activation,
        examples/benchmarks/ResNet.py:65-123): bring x to the waterline and re-encrypt it here -- where the whole layer is one ciphertext --
        unless it still has `need` primes to spend; without hints the lazy policy bootstraps wherever a value runs out, typically inside the
        next convolution where dozens of rotated copies are alive"""
        if not self.lazy:
            return x
        x = self._prepare(x, 0)
        return self._boot(x) if x.level - need < self.min_level else x

    def finish(self, x: Value) -> Value:
        """a program result: rescaled to the waterline like any other consumer would see it"""
        if not self.lazy:
            return x
        while x.scale_bits - self.rescale_bits >= self.waterline and x.level > 1:
            x = self.rescale(x)
        return x

    def _match(self, x: Value, y: Value, scales: bool):
        if self.lazy and scales and x.scale_bits != y.scale_bits:
            # bring the larger scale down by whole primes first; only the remainder is paid for with an upscale
            if x.scale_bits < y.scale_bits:
                y, x = self._match(y, x, True)
                return x, y
            while x.scale_bits - self.rescale_bits >= y.scale_bits:
                x = self._rescale_or_boot(x)
        lv = min(x.level, y.level)
        if scales and x.scale_bits != y.scale_bits:
            s = max(x.scale_bits, y.scale_bits)
            if self.lazy and not self._fits(lv, s):
                x = x if self._fits(x.level, s) else self._boot(x)
                y = y if self._fits(y.level, s) else self._boot(y)
                lv = min(x.level, y.level)
        x, y = self.modswitch(x, x.level - lv), self.modswitch(y, y.level - lv)
        if scales and x.scale_bits != y.scale_bits:
            s = max(x.scale_bits, y.scale_bits)
            x, y = self.upscale(x, s - x.scale_bits), self.upscale(y, s - y.scale_bits)
        return x, y

    def add(self, x: Value, y: Value) -> Value:
        x, y = self._match(x, y, True)
        out = self._new(x.level, y.scale_bits, x.plain + y.plain if self.shadow else None)
        self._emit(OP_ADDCC, out, x, y.id, True)
        return out

    def sub(self, x: Value, y: Value) -> Value:
        return self.add(x, self.negate(y))

    def add_plain(self, x: Value, vec) -> Value:
        reg = self._encode(self._const(vec), x.level, x.scale_bits)
        out = self._new(x.level, x.scale_bits, x.plain + self._tile(vec) if self.shadow else None)
        self._emit(OP_ADDCP, out, x, reg)
        return out

    def mul_plain(self, x: Value, vec, scale_bits=None, normalise=True) -> Value:
        # lazy policy: ct at the waterline times a plaintext at one prime's worth of scale -> one rescale restores it
        sb = (self.rescale_bits if self.lazy else self.waterline) if scale_bits is None else scale_bits
        if self.lazy and self.carry_scale and scale_bits is None:
            while x.scale_bits - self.rescale_bits >= self.waterline:
                x = self._rescale_or_boot(x)
            # land the product on waterline + one prime (or + 20 more bits) with a plaintext of at least waterline bits
            sb = self.waterline + self.rescale_bits - x.scale_bits
            if sb < self.waterline:
                sb += 20
        if self.lazy:
            x = self._prepare(x, sb)
        reg = self._encode(self._const(vec), x.level, sb)
        out = self._new(x.level, x.scale_bits + sb, x.plain * self._tile(vec) if self.shadow else None)
        self._emit(OP_MULCP, out, x, reg)
        return self._normalise(out) if (normalise and not self.lazy) else out

    def normalise(self, x: Value) -> Value:
        """rescale while the scale stays at or above the waterline (what WaterlineRescaling does after a sum)"""
        return self._normalise(x)

    def mul(self, x: Value, y: Value) -> Value:
        if self.lazy:
            same = x is y
            x = self._prepare(x, 0)
            y = x if same else self._prepare(y, 0)
            need = x.scale_bits + y.scale_bits  # only an operand that is itself too low is bootstrapped
            if not self._fits(x.level, need):
                x = self._boot(x)
            y = x if same else (y if self._fits(y.level, need) else self._boot(y))
        x, y = self._match(x, y, False)
        out = self._new(x.level, x.scale_bits + y.scale_bits, x.plain * y.plain if self.shadow else None)
        self._emit(OP_MULCC, out, x, y.id, True)
        return out if self.lazy else self._normalise(out)

    # ---- register allocation + serialisation -----------------------------------------------------------------------
    def assemble(self, preserve_args=True):
        """Returns (cst_bytes, hevm_bytes, info).  Cipher registers: arguments first, then greedy reuse of dead ones.
        preserve_args keeps the argument registers out of the recycling pool, so a loaded program can be run()
        repeatedly on the same encrypted inputs (every other register is written before it is read)."""
        reg_of: dict[int, int] = {}
        free: list[int] = []
        next_reg = 0
        for a in self.args:
            reg_of[a.id] = next_reg
            next_reg += 1
        wire = np.zeros((len(self.ops), 4), dtype=np.uint16)
        for idx, op in enumerate(self.ops):
            if op.opcode in (OP_ENCODE, OP_ENCODE_COMPLEX):
                wire[idx] = (op.opcode, op.dst, op.lhs, op.rhs)
                continue
            lhs_reg = reg_of[op.lhs]
            rhs = reg_of[op.rhs] if op.rhs_is_value else op.rhs
            # sources that die here can be recycled as the destination (kernels are alias-safe)
            dying = [v for v in ({op.lhs} | ({op.rhs} if op.rhs_is_value else set())) if self.values[v].last_use == idx]
            for v in dying:
                if not (preserve_args and reg_of[v] < len(self.args)):
                    free.append(reg_of[v])
            if free:
                dst_reg = free.pop()
            else:
                dst_reg = next_reg
                next_reg += 1
            reg_of[op.dst] = dst_reg
            if self.values[op.dst].last_use < 0:  # never read: dead immediately
                free.append(dst_reg)
            assert max(dst_reg, lhs_reg) < 0xFFFF
            wire[idx] = (op.opcode, dst_reg, lhs_reg, rhs & 0xFFFF)
        hevm = pack_hevm([a.scale_bits for a in self.args], [a.level for a in self.args],
                         [r.scale_bits for r in self.results], [r.level for r in self.results],
                         [reg_of[r.id] for r in self.results], next_reg, self.num_plain, self.init_level, wire)
        info = {"num_ctxt": next_reg, "num_ptxt": self.num_plain, "num_ops": len(self.ops),
                "op_mix": {OP_NAMES[k]: int((wire[:, 0] == k).sum()) for k in range(11)}}
        return pack_cst(self.constants), hevm, info

    def write(self, cst_path, hevm_path):
        cst, hevm, info = self.assemble()
        Path(cst_path).write_bytes(cst)
        Path(hevm_path).write_bytes(hevm)
        return info

    def expected(self):
        return [r.plain for r in self.results]


# ---- programs of the reference's example suite, hand-assembled ------------------------------------------------------
def sobel_filter(image64: np.ndarray, **kw) -> Builder:
    """examples/benchmarks/SobelFilter.py:9-26 (expected result: examples/tests/SobelFilter.py:17-35)."""
    b = Builder(**kw)
    x = b.input(image64)
    F = [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]
    Ix = Iy = None
    for i in range(3):
        for j in range(3):
            rot = b.rotate(x, i * 64 + j)
            for coef, which in ((F[i][j], "x"), (F[j][i], "y")):
                if coef == 0:
                    continue
                h = b.mul_plain(rot, [float(coef)])
                if which == "x":
                    Ix = h if Ix is None else b.add(Ix, h)
                else:
                    Iy = h if Iy is None else b.add(Iy, h)
    c = b.add(b.mul(Ix, Ix), b.mul(Iy, Iy))
    c2 = b.mul(c, c)
    c3 = b.mul(c2, c)
    d = b.add(b.sub(b.mul_plain(c3, [0.173]), b.mul_plain(c2, [1.098])), b.mul_plain(c, [2.214]))
    b.output(d)
    return b


def linear_regression(xs: np.ndarray, ys: np.ndarray, epochs=2, lr=0.01, logn_data=12, **kw) -> Builder:
    """examples/benchmarks/LinearRegression.py:5-36: gradient descent with rotate-and-add reductions."""
    b = Builder(**kw)
    x, y = b.input(xs), b.input(ys)
    n = 1 << logn_data

    def reduce_sum(v):
        for k in range(logn_data):
            v = b.add(v, b.rotate(v, 1 << k))
        return v

    w = bias = None
    for _ in range(epochs):
        pred = x if w is None else b.mul(x, w)
        if w is None:
            pred = b.mul_plain(x, [0.0])
        if bias is not None:
            pred = b.add(pred, bias)
        err = b.sub(pred, y)
        gw = b.mul_plain(reduce_sum(b.mul(err, x)), [lr * 2.0 / n])
        gb = b.mul_plain(reduce_sum(err), [lr * 2.0 / n])
        w = b.negate(gw) if w is None else b.sub(w, gw)
        bias = b.negate(gb) if bias is None else b.sub(bias, gb)
    b.output(w)
    b.output(bias)
    return b


def offset_with_naf_weight(rng, weight: int, max_bit=12) -> int:
    """A rotation offset whose non-adjacent form has exactly `weight` non-zero digits, i.e. that costs `weight`
    key-switch hops under SEAL's default power-of-two Galois keys (Evaluator::rotate_internal)."""
    while True:
        bits = sorted(int(v) for v in rng.choice(np.arange(0, max_bit + 1), size=weight, replace=False))
        if all(b2 - b1 >= 2 for b1, b2 in zip(bits, bits[1:])):
            break
    signs = rng.choice([-1, 1], size=weight)
    off = int(sum(int(sg) * (1 << b) for sg, b in zip(signs, bits)))
    return off if off != 0 else 1


# hop histogram of the traced ResNet: 1 hop 1 769, 2: 286, 3: 185, 4: 144, 5: 66, 6: 18 of 2 510
RESNET_TAP_HOPS = [1] * 75 + [2] * 14 + [3] * 9 + [4] * 7 + [5] * 3 + [6] * 1


def resnet_shaped(seed=100, slots=1 << 14, layers=20, init_level=13, waterline=40, shadow=False, mulcp_per_tap=2,
                  mulcc_per_layer=18, addcp_per_layer=30, neg_per_layer=7, tap_hops=None, reduce_steps=13, conv_level=3,
                  act_level=5, min_level=2) -> Builder:
    """A synthetic program with the traced op mix of examples/benchmarks/ResNet.py at nt = 2^14 (SURVEY.md App. C:
    2 510 rotates = 3 910 key-switch hops, 4 822 ct*pt, 361 ct*ct, ~5.9 k ct+ct, 591 ct+pt, 133 negates).

    Level schedule: the reference's own numbers (README.md:176-188: 53.7 s over ~4.3 k key switches = 12.6 ms each,
    against profiled_SEAL_CPU.json's 8.0 ms at 2 primes / 13.6 ms at 3) say the DaCapo-compiled program keeps its
    ciphertexts at 2-3 primes -- cheap because the SEAL VM's "bootstrap" is a decrypt/re-encrypt -- so every layer
    here runs its convolution rotations at `conv_level` primes and its ct*ct chain between `act_level` and
    `min_level` (= the config's levelLowerBound 2), re-encrypting (opcode 10) whenever it runs out of levels.
    Each "layer" = multiplexed convolution (rotate, weight multiply, accumulate), rotate-and-sum channel reduction,
    bias adds, polynomial activation.  No compiled ResNet .hevm can be produced here (needs hecate-opt); if one is
    supplied the runtime runs it unchanged."""
    rng = np.random.default_rng(seed)
    b = Builder(slots=slots, waterline=waterline, init_level=init_level, shadow=shadow, min_level=1)
    x = b.input(rng.uniform(-0.5, 0.5, slots) if shadow else None)
    tap_hops = RESNET_TAP_HOPS if tap_hops is None else tap_hops
    max_bit = int(np.log2(slots)) - 2
    w = lambda s=0.1: rng.uniform(-s, s, slots if shadow else 16)  # noqa: E731

    def at_level(v, lvl):  # bring a value to exactly `lvl` primes
        if v.level < lvl:
            return b.bootstrap(v, lvl)
        return b.modswitch(v, v.level - lvl)

    for layer in range(layers):
        x = at_level(x, conv_level)
        acc = None
        for h in tap_hops:
            r = b.rotate(x, offset_with_naf_weight(rng, h, max_bit))
            for _ in range(mulcp_per_tap):  # products are summed first and rescaled once, as the compiler schedules it
                term = b.mul_plain(r, w(0.02), normalise=False)
                acc = term if acc is None else b.add(acc, term)
        acc = b.normalise(acc)
        acc = b.normalise(b.mul_plain(acc, [2.0 ** -min(reduce_steps, 6)], scale_bits=60 - acc.scale_bits if acc.scale_bits < 60
                                      else 20, normalise=False))
        for k in range(reduce_steps):  # rotate-and-sum reduction across packed channels
            acc = b.add(acc, b.rotate(acc, 1 << (k % (max_bit + 1))))
        for _ in range(addcp_per_layer):
            acc = b.add_plain(acc, w(0.01))
        for _ in range(neg_per_layer // 2):
            acc = b.negate(b.negate(acc))
        if neg_per_layer % 2:
            acc = b.negate(acc)
        # activation: a chain of mulcc_per_layer ct*ct products, one level each
        t = at_level(acc, act_level)
        base = {act_level: t}
        for i in range(mulcc_per_layer):
            if t.level <= min_level:
                t = b.bootstrap(t, act_level)
            if t.level not in base:
                base[t.level] = b.modswitch(base[act_level], act_level - t.level)
            t = b.mul(t, base[t.level])
            if i % 6 == 5 and t.level > min_level:
                t = b.mul_plain(t, w(0.5), scale_bits=60)
        x = t
    b.output(x)
    return b
