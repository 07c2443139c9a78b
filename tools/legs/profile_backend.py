#!/usr/bin/env python3
"""Per-op, per-level latency table of this runtime in the format the reference's compiler reads with
`--ckks-config=profiled_<lib>_<hw>.json` (/root/reference/lib/Dialect/Earth/IR/EarthDialect.cpp:134-181; the shipped
tables are profiled_SEAL_CPU.json, profiled_HEAAN_GPU.json): microseconds, list index 0 = 1 prime.  DaCapo's bootstrap
placement and the scale-management passes minimise the sum of these numbers, so this file is what lets the reference's
compiler plan FOR the MI355X backend (e.g. opcode 10 is not free here, and a key switch at 1 prime costs 70 % of one at 4).

What is measured is the latency of one op inside a dependent chain through run() (batched plan, one ciphertext): the
cost a sequential program section pays.  Large independent batches cost far less per op (throughput mode).

    python tools/legs/profile_backend.py [--out profiles/r01_profiled_SEAL_MI355X.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import runner  # noqa: E402

MAXL = 13


def run_time(hevm, b, reps=4):
    cst, hv, _ = b.assemble()
    hevm.load_mem(cst, hv)
    for i in range(hevm.arglen):
        hevm.setInput(i, np.linspace(-0.5, 0.5, hevm.slots) * (i + 1))
    hevm.run()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        hevm.run()
        best = min(best, time.perf_counter() - t0)
    return best


def chain(hevm, level, n, body, inputs=1):
    """time per iteration (us) of x = body(b, x, others) repeated n times at `level` primes"""
    b = ha.Builder(slots=hevm.slots, init_level=level, shadow=False)
    xs = [b.input(None) for _ in range(inputs)]
    x = xs[0]
    for _ in range(n):
        x = body(b, x, xs)
    b.output(x)
    return run_time(hevm, b) / n * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(Path(__file__).resolve().parents[2] / "profiles" / "r01_profiled_SEAL_MI355X.json"))
    a = ap.parse_args()
    hevm = runner.HEVM(fresh=True, logN=15, num_primes=14)
    empty = chain(hevm, 2, 1, lambda b, x, xs: b.negate(x)) * 0  # warm
    lat = {k: [] for k in ("earth.rotate_single", "earth.rescale_single", "earth.modswitch_single", "earth.add_single", "earth.add_double",
                           "earth.mul_single", "earth.mul_double", "earth.negate_single", "earth.bootstrap_single")}
    for lvl in range(1, MAXL + 1):
        neg = chain(hevm, lvl, 200, lambda b, x, xs: b.negate(x))
        lat["earth.negate_single"].append(neg)
        lat["earth.add_single"].append(chain(hevm, lvl, 200, lambda b, x, xs: b.add_plain(x, [0.001])))
        lat["earth.add_double"].append(chain(hevm, lvl, 200, lambda b, x, xs: b.add(x, xs[1]), inputs=2))
        lat["earth.mul_single"].append(chain(hevm, lvl, 16, lambda b, x, xs: b.mul_plain(x, [0.999], normalise=False)))
        lat["earth.mul_double"].append(chain(hevm, lvl, 16, lambda b, x, xs: _mul_raw(b, x, xs[1]), inputs=2))
        lat["earth.rotate_single"].append(chain(hevm, lvl, 100, lambda b, x, xs: b.rotate(x, 1)))
        lat["earth.modswitch_single"].append(0.0)  # a view of the same buffer in the execution plan
        # opcode 10 INTO this level (the table is indexed by the level the result has): negate keeps the chain dependent
        boot = chain(hevm, lvl, 60, lambda b, x, xs, t=lvl: b.bootstrap(b.negate(x), t)) - neg
        lat["earth.bootstrap_single"].append(boot)
        if lvl >= 2:  # [upscale, rescale, opcode 10 back to lvl] minus its other two members
            with_rs = chain(hevm, lvl, 40, lambda b, x, xs, t=lvl: b.bootstrap(b.rescale(b.upscale(x, 60)), t))
            boot_low = chain(hevm, lvl, 60, lambda b, x, xs, t=lvl: b.bootstrap(b.modswitch(b.negate(x), 1), t)) - neg
            lat["earth.rescale_single"].append(max(with_rs - boot_low - lat["earth.mul_single"][-1], 0.0))
        else:
            lat["earth.rescale_single"].append(0.0)
        print(f"level {lvl:2d}: " + "  ".join(f"{k.split('.')[1]} {v[-1]:7.1f}" for k, v in lat.items()), flush=True)
    del empty
    table = {k: [int(round(x)) for x in v] for k, v in lat.items()}
    out = {
        "runtime": "SEAL-HEVM", "rescalingFactor": 60, "polynomialDegree": 32768, "levelLowerBound": 1, "levelUpperBound": MAXL,
        "bootstrapLevelLowerBound": 1, "bootstrapLevelUpperBound": MAXL,
        "latencyTable": table,
        "latencyTable_us_float": {k: [round(x, 2) for x in v] for k, v in lat.items()},
        "noiseTable": {},
        "_comment": "latency of one op in a dependent chain through run() on one MI355X (tools/legs/profile_backend.py); same arithmetic "
                    "as SEAL, so profiled_SEAL_CPU.json's noiseTable applies unchanged",
    }
    Path(a.out).write_text(json.dumps(out, indent=1))
    print("wrote", a.out)


def _mul_raw(b, x, y):
    """ct*ct without the builder's eager rescale (the chain stays at one level; only the bookkeeping scale grows)"""
    out = b._new(x.level, x.scale_bits + y.scale_bits, None)
    b._emit(ha.OP_MULCC, out, x, y.id, True)
    return out


if __name__ == "__main__":
    main()
