#!/usr/bin/env python3
"""Latency of dependent op chains through run() (N = 2^15, 14 primes): what one step of a sequential program section
costs on the MI355X.  usage: python tools/legs/chain_bench.py"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import runner  # noqa: E402


def timed(hevm, b, reps=5):
    cst, hv, info = b.assemble()
    hevm.load_mem(cst, hv)
    hevm.setInput(0, np.linspace(-0.5, 0.5, hevm.slots))
    hevm.run()
    t0 = time.perf_counter()
    for _ in range(reps):
        hevm.run()
    return (time.perf_counter() - t0) / reps, info


def main():
    hevm = runner.HEVM(fresh=True, logN=15, num_primes=14)
    rows = []
    for lvl in (1, 2, 3, 4):
        n = 120
        b = ha.Builder(slots=hevm.slots, init_level=lvl, shadow=False)
        x = b.input(None)
        for _ in range(n):
            x = b.rotate(x, 1)
        b.output(x)
        t, _ = timed(hevm, b)
        rows.append((f"rotate by 1 (1 hop), level {lvl}", n, t))
    for lvl in (2, 3, 4):
        n = 200
        b = ha.Builder(slots=hevm.slots, init_level=lvl, shadow=False)
        x = b.input(None)
        for k in range(n):
            x = b.add_plain(x, [0.001 * (k % 7)]) if k % 2 else b.negate(x)
        b.output(x)
        t, _ = timed(hevm, b)
        rows.append((f"addcp / negate alternating, level {lvl}", n, t))
    n = 60
    b = ha.Builder(slots=hevm.slots, init_level=3, policy="lazy", boot_level=3, shadow=False)
    x = b.input(None)
    for _ in range(n):
        x = b.mul(x, x)
    b.output(b.finish(x))
    t, info = timed(hevm, b)
    rows.append((f"x = x*x chain, lazy policy, boot 3 {info['op_mix']}", n, t))
    for lvl in (2, 3):
        n = 100
        b = ha.Builder(slots=hevm.slots, init_level=lvl, shadow=False)
        x = b.input(None)
        for _ in range(n):
            x = b.bootstrap(b.negate(x), lvl)
        b.output(x)
        t, _ = timed(hevm, b)
        rows.append((f"negate + opcode 10 to level {lvl}", n, t))
    for name, n, t in rows:
        print(f"{name:80s} {t*1e3:8.2f} ms  {t/n*1e6:8.1f} us/op")


if __name__ == "__main__":
    main()
