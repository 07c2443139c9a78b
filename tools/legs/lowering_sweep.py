#!/usr/bin/env python3
"""The three lowerings of the ResNet-20 trace (opcode 10 -> 3 / 6 / 13 primes) under a list of launch-shape option sets, one VM:
    python tools/legs/lowering_sweep.py [steps=5] [--only b13] [--new-vm] "name=value,name=value" ...
Launch shapes are read when a launch is issued, i.e. when load() records the plan's graph: the program is re-loaded per option set.
Prints best-of-steps ms per run() for each lowering; the first row is the defaults."""
import gzip
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import runner  # noqa: E402

args = sys.argv[1:]
steps = 5
if args and args[0].isdigit():
    steps = int(args.pop(0))
tags = ["", ".b6", ".b13"]
if "--only" in args:
    i = args.index("--only")
    tags = ["" if args[i + 1] in ("b3", "headline") else "." + args[i + 1]]
    del args[i:i + 2]
fx = ha.read_fixture(ROOT / "tests" / "golden" / "resnet20")
progs = {t: (fx["hevm"] if not t else gzip.open(ROOT / "tests" / "golden" / f"resnet20{t}.hevm.gz").read()) for t in tags}
new_vm = "--new-vm" in args  # VM options (max_batch, plan_lanes ...) are read when a VM is created: one VM per option set
args = [a for a in args if a != "--new-vm"]
vm = None if new_vm else runner.HEVM(fresh=True, logN=15, num_primes=14)
for spec in [""] + args:
    runner.reinit_lw().hevm_reset_options()
    for kv in filter(None, spec.split(",")):
        k, _, v = kv.partition("=")
        runner.set_option(k, int(v, 0))
    if new_vm:
        if vm is not None:
            vm.close()
        vm = runner.HEVM(fresh=True, logN=15, num_primes=14)
    row = []
    for t in tags:
        vm.load_mem(fx["cst"], progs[t])
        vm.setInput(0, fx["packed"])
        vm.run()
        best = 1e9
        for _ in range(steps):
            t0 = time.perf_counter()
            vm.run()
            best = min(best, time.perf_counter() - t0)
        row.append(f"{'b3' if not t else t[1:]} {best * 1e3:8.2f} ms")
    print(f"{spec or 'defaults':60s} " + "   ".join(row), flush=True)
runner.lw.hevm_reset_options()
