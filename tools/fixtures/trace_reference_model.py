#!/usr/bin/env python
"""Trace one of the reference's model scripts (examples/benchmarks/*.py) into an HEVM program WITHOUT its MLIR stack.

Runs only in the build container (it imports /root/reference/python/poly and executes the benchmark script there);
nothing here is used at run time on the GPU box.  What it produces is DATA: the op stream of the model as the
reference's own frontend code emits it (rotation offsets, ct*pt / ct*ct / add structure, constant lengths), lowered by
`dacapo_amd.hevm_asm.Builder` (policy="lazy": rescale-on-demand + automatic bootstrap placement) to `.hevm` bytecode.

    python tools/fixtures/trace_reference_model.py ResNet --out tests/golden/resnet20

writes  <out>.hevm.gz    the bytecode (HEVMHeader.h wire format, gzip)
        <out>.cst.xz     the constant file (ElideConstant.cpp:40-53 format, xz): BN-folded weights in packed slot layout
        <out>.input.npz  packed input image, the torch model's logits for it, plaintext evaluation of the program
        <out>.json       op mix, level histogram, NTT-equivalents, provenance
and with --full also <out>.cst uncompressed (~0.5 GB -- never committed).

The `hecate` module the scripts import is replaced by a shim with the same surface as
/root/reference/python/hecate/hecate/expr.py (Expr with + - * neg rotate, Plain, Empty, func, save, bootstrap);
`torchvision` (absent in this image, imported but unused by the benchmark scripts) is stubbed.
"""
from __future__ import annotations

import argparse
import gzip
import hashlib
import json
import lzma
import os
import sys
import types
from collections.abc import Iterable
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from dacapo_amd import hevm_asm  # noqa: E402

REF = Path("/root/reference")


def make_shim(builder: hevm_asm.Builder, inputs):
    """A stand-in for the `hecate` frontend module that lowers straight into `builder`."""
    import torch

    hc = types.ModuleType("hecate")
    state = {"funcs": [], "hints": 0}

    class Expr:
        __array_priority__ = 1000  # numpy defers to our reflected operators

        def __init__(self, ct=None, pt=None):
            self.ct, self.pt = ct, pt

        # expr.py:147-175 -- note __i<op>__ is bound to the REVERSED method there (a -= b yields b - a); kept.
        def __add__(self, other):
            return _binary("add", self, resolve(other))

        def __radd__(self, other):
            return _binary("add", resolve(other), self)

        __iadd__ = __radd__

        def __sub__(self, other):
            return _binary("sub", self, resolve(other))

        def __rsub__(self, other):
            return _binary("sub", resolve(other), self)

        __isub__ = __rsub__

        def __mul__(self, other):
            return _binary("mul", self, resolve(other))

        def __rmul__(self, other):
            return _binary("mul", resolve(other), self)

        __imul__ = __rmul__

        def __neg__(self):
            if self.ct is None:
                return Expr(pt=-self.pt)
            return Expr(ct=builder.negate(self.ct))

        def rotate(self, offset):
            offset = int(offset)
            if self.ct is None:
                return Expr(pt=np.roll(_tile(self.pt), -offset))
            if offset % builder.slots == 0:
                return self
            return Expr(ct=builder.rotate(self.ct, _wrap(offset)))

        def __copy__(self):
            raise Exception("Copying Hecate object is forbidden")

        __deepcopy__ = __copy__

    def _wrap(offset):  # the wire format carries int16 offsets (CKKSOps.td rotate); the model's own offsets fit
        if -(1 << 15) <= offset < (1 << 15):
            return offset
        offset %= builder.slots
        return offset - builder.slots if offset > builder.slots // 2 else offset

    def _tile(vec):
        vec = np.asarray(vec, dtype=np.float64).ravel()
        return vec[np.arange(builder.slots) % len(vec)]

    class Plain(Expr):
        def __init__(self, data, scale=40):
            super().__init__(pt=np.array(np.asarray(data, dtype=np.float64).ravel().tolist(), dtype=np.float64))

    def resolve(other):  # expr.py:236-252
        if isinstance(other, Expr):
            return other
        if isinstance(other, (int, float, np.integer, np.floating)):
            return Plain(np.array([float(other)]))
        if isinstance(other, list):
            return Plain(np.array(other, dtype=np.float64))
        if isinstance(other, torch.Tensor):
            return Plain(torch.flatten(other).tolist())
        if isinstance(other, np.ndarray):
            return Plain(other)
        raise Exception(f"Cannot create compatible type from {type(other)}")

    def _pp(op, a, b):
        if len(a) != len(b) and len(a) != 1 and len(b) != 1:
            a, b = _tile(a), _tile(b)
        return {"add": a + b, "sub": a - b, "mul": a * b}[op]

    def _binary(op, x: Expr, y: Expr) -> Expr:
        if x.ct is None and y.ct is None:
            return Expr(pt=_pp(op, x.pt, y.pt))
        if x.ct is not None and y.ct is not None:
            return Expr(ct={"add": builder.add, "sub": builder.sub, "mul": builder.mul}[op](x.ct, y.ct))
        if x.ct is not None:  # ct (op) pt
            if op == "add":
                return Expr(ct=builder.add_plain(x.ct, y.pt))
            if op == "sub":
                return Expr(ct=builder.add_plain(x.ct, -y.pt))
            return Expr(ct=builder.mul_plain(x.ct, y.pt))
        # pt (op) ct
        if op == "add":
            return Expr(ct=builder.add_plain(y.ct, x.pt))
        if op == "sub":
            return Expr(ct=builder.add_plain(builder.negate(y.ct), x.pt))
        return Expr(ct=builder.mul_plain(y.ct, x.pt))

    class Empty:  # expr.py:272-287
        def __add__(self, other):
            return resolve(other)

        __radd__ = __iadd__ = __sub__ = __rsub__ = __isub__ = __add__

    def bootstrap(x):
        # expr.py:112-126.  On an iterable the reference rebinds a loop variable and returns its argument unchanged;
        # on a single Expr it inserts a bootstrap HINT.  The DaCapo pipeline decides the real placement itself, and so
        # does Builder(policy="lazy"); the hints are counted and dropped.
        state["hints"] += 1
        need = getattr(builder, "hint_need", None)
        if need is None:
            return x
        # with --hint-need the hints are honoured: every ciphertext of the layer is re-encrypted here if it cannot pay for `need` more primes
        if isinstance(x, np.ndarray):  # the layers hand object arrays of ciphertexts around (element-wise operators)
            out = np.empty_like(x)
            for idx, e in np.ndenumerate(x):
                out[idx] = Expr(ct=builder.hint(e.ct, need)) if getattr(e, "ct", None) is not None else e
            return out
        if isinstance(x, Iterable):
            return type(x)(Expr(ct=builder.hint(e.ct, need)) if getattr(e, "ct", None) is not None else e for e in x)
        return Expr(ct=builder.hint(x.ct, need)) if x.ct is not None else x

    class Func:
        def __init__(self, fun, paramstr):
            self.fun, self.params = fun, paramstr.split(",")

        def eval(self):
            args = []
            for k, kind in enumerate(self.params):
                assert kind.strip() == "c"
                args.append(Expr(ct=builder.input(inputs[k] if k < len(inputs) else None)))
            ret = self.fun(*args)
            if isinstance(ret, Expr) or not isinstance(ret, Iterable):
                ret = [ret]
            for r in ret:
                builder.output(builder.finish(r.ct))

    def func(param):
        def gen(f):
            fn = Func(f, param)
            state["funcs"].append(fn)
            return fn
        return gen

    def save(dirs="", cst_dirs=""):
        for f in state["funcs"]:
            f.eval()
        return "traced"

    class HEVM:
        """stand-in for python/hecate/hecate/runner.py:174-271 used when the reference's examples/tests/<name>.py is run
        against a traced program: inputs are recorded, run() evaluates the program on cleartext (hevm_asm.plain_eval)"""

        def __init__(self, *a, **k):
            self.inputs, self.out, self.rms, self.hevm_path = {}, None, None, ""

        def load(self, cst_path, hevm_path):
            self.hevm_path = hevm_path

        def setInput(self, i, data):
            self.inputs[i] = np.asarray(data, dtype=np.float64).ravel().copy()

        def run(self):
            prog = state["program"]
            self.out = np.stack(hevm_asm.plain_eval(prog[1], prog[0], [self.inputs[i] for i in sorted(self.inputs)], builder.slots))

        def getOutput(self):
            return self.out

        def printer(self, latency, rms, mem_usage=0.0):
            self.rms = float(np.max(rms)) if np.ndim(rms) else float(rms)
            state["runner"] = self

        def setDebug(self, enable):
            pass

    hc.HEVM = HEVM
    hc.setLibnHW = lambda argv: None
    hc.Expr, hc.Plain, hc.Empty, hc.func, hc.save, hc.bootstrap = Expr, Plain, Empty, func, save, bootstrap
    hc.hecate_dir = str(REF)
    hc._state, hc._inputs = state, inputs
    return hc


def stub_torchvision():
    tv = types.ModuleType("torchvision")
    for sub in ("transforms", "datasets"):
        m = types.ModuleType(f"torchvision.{sub}")
        setattr(tv, sub, m)
        sys.modules[f"torchvision.{sub}"] = m
    sys.modules["torchvision"] = tv


MODELS = {  # script name -> (module under poly.models, constructor, weights file, key of the state dict, first convolution)
    "ResNet": ("ResNet", "resnet20", "resnet20.silu.model", "state_dict", lambda m: m.conv1),
    # SqueezeNet ("SqueezeNet", "squeezenet", "squeezeNet_silu_avgpool_model", None, m.conv_1.Conv2d) traces, but on a synthetic
    # image its activations leave the interval of the SiLU polynomial (values ~1e8): it needs real CIFAR-10 inputs
}


def model_input(a, slots):
    """A CIFAR-shaped synthetic image (no dataset in this image): smooth random field in [0,1], normalised and packed
    exactly as examples/tests/<model>.py does (ResNet.py:31,50-69), plus the torch model's own answer for it."""
    import importlib

    import torch
    from poly.MPCB import CascadeConv, shapeClosure
    mod, ctor, weights, key, first_conv = MODELS[a.model]
    rng = np.random.default_rng(a.seed)
    coarse = rng.uniform(0.0, 1.0, (3, 6, 6))
    img = torch.nn.functional.interpolate(torch.tensor(coarse)[None], size=(32, 32), mode="bicubic", align_corners=False)[0]
    img = img.clamp(0.0, 1.0)
    mean = torch.tensor([0.485, 0.456, 0.406])[:, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    x = ((img - mean) / std)[None].double()
    model = torch.nn.DataParallel(getattr(importlib.import_module(f"poly.models.{mod}"), ctor)())
    sd = torch.load(str(REF / "examples/data" / weights), map_location="cpu")
    if key:
        model.load_state_dict(sd[key])
    else:
        model.module.load_state_dict(sd)
    model = model.eval().double().cpu()
    with torch.no_grad():
        torch_res = model.module(x).numpy()[0]
    shapes = CascadeConv({"nt": slots, "bb": 32, "ko": 1, "ho": 32, "wo": 32}, first_conv(model.module))
    packed = np.asarray(shapeClosure(**shapes)["MPP"](x)[0], dtype=np.float64).ravel()
    return x.numpy(), packed, torch_res


def level_histogram(b: hevm_asm.Builder):
    """key-switch hops per level (SEAL NAF hop count per rotate, 1 per mulcc)"""
    def naf_weight(k):
        k = abs(k)
        w = 0
        while k:
            if k & 1:
                d = 2 - (k & 3)
                k -= d
                w += 1
            k >>= 1
        return w
    hops, muls, boots = {}, {}, 0
    for op in b.ops:
        if op.opcode == hevm_asm.OP_ROTATE:
            off = op.rhs - 65536 if op.rhs >= 32768 else op.rhs
            lvl = b.values[op.lhs].level
            # SEAL rotates by the NAF of the step reduced to (-slots/2, slots/2]
            s = off % b.slots
            s = s - b.slots if s > b.slots // 2 else s
            hops[lvl] = hops.get(lvl, 0) + naf_weight(s)
        elif op.opcode == hevm_asm.OP_MULCC:
            lvl = b.values[op.lhs].level
            muls[lvl] = muls.get(lvl, 0) + 1
        elif op.opcode == hevm_asm.OP_BOOTSTRAP:
            boots += 1
    return hops, muls, boots


def ntt_equivalents(b: hevm_asm.Builder):
    """SURVEY.md s6 cost model: key switch at l primes = (l+1)(l+2) NTT-equivalents, rescale = 2l, opcode 10 =
    decrypt (l) + encode (target) + zero-encryption at target+1 primes (2(t+1) forward + 2(t+1) inverse... counted 5t+4)"""
    hops, muls, _ = level_histogram(b)
    total = sum((l + 1) * (l + 2) * n for l, n in hops.items()) + sum((l + 1) * (l + 2) * n for l, n in muls.items())
    for op in b.ops:
        if op.opcode == hevm_asm.OP_RESCALE:
            total += 2 * b.values[op.lhs].level
        elif op.opcode == hevm_asm.OP_BOOTSTRAP:
            total += b.values[op.lhs].level + 5 * op.rhs + 4
    return total


SUITE = ["SobelFilter", "HarrisCornerDetection", "LinearRegression", "PolynomialRegression", "Multivariate", "MLP"]


def trace_suite_program(name, slots_log, waterline, boot_level, out_dir):
    """One of the non-ResNet benchmarks: trace examples/benchmarks/<name>.py (inputs unknown at trace time), give the
    program the fewest primes it needs (what the reference's level assignment does), then run the reference's own
    examples/tests/<name>.py against the traced program evaluated on cleartext to record its inputs, the expected
    outputs and the error figure the script prints."""
    slots = 1 << slots_log
    script = REF / "examples/benchmarks" / f"{name}.py"
    test = REF / "examples/tests" / f"{name}.py"

    def trace(init_level):
        b = hevm_asm.Builder(slots=slots, waterline=waterline, init_level=init_level, policy="lazy", boot_level=boot_level,
                             shadow=False)
        for m in [k for k in sys.modules if k == "hecate" or k.startswith("poly")]:
            del sys.modules[m]
        sys.modules["hecate"] = make_shim(b, [])
        exec(compile(script.read_text(), str(script), "exec"), {"__name__": "__main__", "__file__": str(script)})
        return b

    b = trace(13)
    if not any(op.opcode == hevm_asm.OP_BOOTSTRAP for op in b.ops):
        slack = min(r.level for r in b.results) - 1
        if slack > 0:
            b = trace(13 - slack)
    cst, hevm, info = b.assemble()
    hc = sys.modules["hecate"]
    hc._state["program"] = (cst, hevm)
    argv, cwd = sys.argv, os.getcwd()
    sys.argv = [str(test), "dacapo", str(waterline), "SEAL", "CPU"]
    try:
        exec(compile(test.read_text(), str(test), "exec"), {"__name__": "__main__", "__file__": str(test)})
    finally:
        sys.argv = argv
        os.chdir(cwd)
    run = hc._state["runner"]
    hops, muls, boots = level_histogram(b)
    meta = {"source": f"examples/benchmarks/{name}.py traced; inputs and error figure from examples/tests/{name}.py run against the "
                      "cleartext evaluation of the traced program (tools/fixtures/trace_reference_model.py --suite)",
            "slots": slots, "waterline": waterline, "init_level": b.init_level, "boot_level": boot_level, "info": info,
            "num_inputs": len(run.inputs), "num_results": len(b.results), "hops_per_level": {str(k): v for k, v in sorted(hops.items())},
            "mulcc_per_level": {str(k): v for k, v in sorted(muls.items())}, "bootstraps": boots,
            "ntt_equivalents": ntt_equivalents(b), "script_rms_on_cleartext": run.rms,
            "hevm_sha256": hashlib.sha256(hevm).hexdigest(), "cst_sha256": hashlib.sha256(cst).hexdigest()}
    out = Path(out_dir) / name
    out.parent.mkdir(parents=True, exist_ok=True)
    with gzip.GzipFile(str(out) + ".hevm.gz", "wb", mtime=0) as f:
        f.write(hevm)
    Path(str(out) + ".cst.xz").write_bytes(lzma.compress(cst, format=lzma.FORMAT_XZ, preset=6))
    np.savez_compressed(str(out) + ".io.npz", expected=run.out, **{f"input{i}": run.inputs[i] for i in sorted(run.inputs)})
    Path(str(out) + ".json").write_text(json.dumps(meta, indent=1))
    print(f"{name:24s} init level {b.init_level:2d}  ops {info['num_ops']:6d}  key switches {sum(hops.values()) + sum(muls.values()):5d}  "
          f"bootstraps {boots:3d}  script rms on cleartext {run.rms:.3e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="ResNet")
    ap.add_argument("--out", default=str(ROOT / "tests/golden/resnet20"))
    ap.add_argument("--slots-log", type=int, default=14)
    ap.add_argument("--boot-level", type=int, default=3,
                    help="primes after opcode 10; 3 is the fastest lowering on the MI355X runtime (2: 1538 bootstraps, 4: +12 %% wall)")
    ap.add_argument("--init-level", type=int, default=3)
    ap.add_argument("--waterline", type=int, default=40)
    ap.add_argument("--rotate-reserve", type=int, default=0)
    ap.add_argument("--headroom", type=int, default=16,
                    help="bits kept free above a value's scale for the magnitude of its slots when deciding how many primes it needs (16 next "
                         "to 60-bit primes; 11 gives 51-bit primes the same level structure: 2 x 51 - 91 = 11 where 2 x 60 - 100 = 20).  The shadow "
                         "evaluation still refuses a value that does not fit its primes")
    ap.add_argument("--rescale-bits", type=int, default=60,
                    help="width of the chain's rescale primes the program's scale management assumes: 60 = the reference's SEAL chain; 51 = the "
                         "HEaaN configuration's rescalingFactor (profiled_HEAAN_GPU.json), for a mixed 60/51-bit chain")
    ap.add_argument("--carry-scale", action="store_true")
    ap.add_argument("--no-shadow", action="store_true")
    ap.add_argument("--full", action="store_true", help="also write the real constants (<out>.cst, not committed)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--real-boot-primes", type=int, default=0,
                    help="K > 0: every bootstrap is REAL CKKS bootstrapping (dacapo_amd/ckks_boot.py) on a chain of K primes "
                         "(K = boot-level + 17 with the default r = 5) instead of opcode 10")
    ap.add_argument("--msg-bits", type=int, default=7, help="--real-boot-primes: bound 2^msg_bits on the magnitude of a bootstrapped value")
    ap.add_argument("--hint-need", type=int, default=None,
                    help="honour the model script's hc.bootstrap hints: bootstrap there unless the value still has this many primes to spend "
                         "(default: hints are dropped and the lazy policy bootstraps wherever a value runs out)")
    ap.add_argument("--suite", action="store_true", help="trace the small benchmarks (%s) into <out>/<name>.*" % ", ".join(SUITE))
    a = ap.parse_args()
    if a.suite:
        stub_torchvision()
        sim = types.ModuleType("simfhe")
        sim.simulate = lambda path: "n/a"
        sys.modules["simfhe"] = sim
        os.environ.setdefault("HECATE", str(REF))
        names = SUITE if a.model == "ResNet" else [a.model]
        for name in names:
            trace_suite_program(name, a.slots_log, a.waterline, a.boot_level, a.out)
        return

    script = REF / "examples/benchmarks" / f"{a.model}.py"
    src = script.read_text()
    # examples/benchmarks/ResNet.py:50-51: the slot count is hard-coded for the HEAAN target (2^16); the SEAL runtime
    # has 2^14 slots (SEAL_HEVM.cpp:39 N = 2^15) and the script carries the alternative as a comment.
    src = src.replace('"nt" : 2**16', f'"nt" : 2**{a.slots_log}')

    slots = 1 << a.slots_log
    b = hevm_asm.Builder(slots=slots, waterline=a.waterline, init_level=a.init_level, policy="lazy", boot_level=a.boot_level, rotate_reserve=a.rotate_reserve, carry_scale=a.carry_scale,
                         rescale_bits=a.rescale_bits, headroom=a.headroom,
                         shadow=not a.no_shadow,
                         real_boot=dict(num_primes=a.real_boot_primes, msg_bits=a.msg_bits) if a.real_boot_primes else None)
    b.hint_need = a.hint_need
    stub_torchvision()
    sys.path.insert(0, str(REF / "python/poly"))
    os.environ.setdefault("HECATE", str(REF))
    sys.modules["hecate"] = make_shim(b, [None])
    image, packed, torch_res = model_input(a, slots)
    sys.modules["hecate"]._inputs[0] = packed
    g = {"__name__": "__main__", "__file__": str(script)}
    exec(compile(src, str(script), "exec"), g)

    cst, hevm, info = b.assemble()
    hops, muls, boots = level_histogram(b)
    lens = [int(len(c)) for c in b.constants]
    meta = {
        "source": f"examples/benchmarks/{a.model}.py traced through python/poly with tools/fixtures/trace_reference_model.py",
        "slots": slots, "waterline": a.waterline, "init_level": a.init_level, "boot_level": a.boot_level, "hint_need": a.hint_need,
        "rescale_bits": a.rescale_bits, "headroom": a.headroom,
        "input": {"packed_len": int(len(packed)), "seed": a.seed, "kind": "smooth synthetic 3x32x32 image, CIFAR-normalised"},
        "torch_result": [float(v) for v in torch_res],
        "info": info,
        "constant_lengths_rle": rle(lens),
        "num_constants": len(lens),
        "hops_per_level": {str(k): v for k, v in sorted(hops.items())},
        "mulcc_per_level": {str(k): v for k, v in sorted(muls.items())},
        "bootstraps": boots,
        "ntt_equivalents": ntt_equivalents(b),
        "bootstrap_hints_dropped": sys.modules["hecate"]._state["hints"] if a.hint_need is None else 0,
        "bootstrap_hints_honoured": sys.modules["hecate"]._state["hints"] if a.hint_need is not None else 0,
        "bootstrapped_value_peaks": sorted(round(v, 3) for v in getattr(b, "boot_peaks", []))[-8:],
        "real_boot": ({k: v for k, v in b.real_boot.items()} if b.real_boot else None),
        "hevm_sha256": hashlib.sha256(hevm).hexdigest(),
        "cst_sha256": hashlib.sha256(cst).hexdigest(),
    }
    if not a.no_shadow:
        exp = b.expected()
        meta["expected"] = [[float(x) for x in e[:16]] for e in exp]
        # examples/tests/ResNet.py:76-81 postprocess: first 10 slots * 32 (HE_Linear is traced with scale = 32)
        got = exp[0][:torch_res.size].reshape(torch_res.shape) * 32
        meta["plain_vs_torch_rms"] = float(np.sqrt(np.mean((got - torch_res) ** 2)))
        print("plaintext evaluation of the traced program vs torch model: rms", meta["plain_vs_torch_rms"])
        print(" traced:", np.round(got, 4), "\n torch :", np.round(torch_res, 4))
    out = Path(a.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    with gzip.GzipFile(str(out) + ".hevm.gz", "wb", mtime=0) as f:
        f.write(hevm)
    Path(str(out) + ".json").write_text(json.dumps(meta, indent=1))
    # the constants are the model's weights replicated over the slots (~14 distinct values per vector): 488 MB -> <2 MB
    Path(str(out) + ".cst.xz").write_bytes(lzma.compress(cst, format=lzma.FORMAT_XZ, preset=6))
    np.savez_compressed(str(out) + ".input.npz", packed=packed, torch_result=torch_res,
                        expected=(b.expected()[0] if not a.no_shadow else np.zeros(0)))
    if a.full:
        Path(str(out) + ".cst").write_bytes(cst)
    print(json.dumps({k: meta[k] for k in ("info", "num_constants", "ntt_equivalents", "hops_per_level", "mulcc_per_level", "bootstraps",
                                           "bootstrap_hints_dropped")}, indent=1))


def rle(xs):
    out = []
    for x in xs:
        if out and out[-1][0] == x:
            out[-1][1] += 1
        else:
            out.append([x, 1])
    return out


if __name__ == "__main__":
    main()
