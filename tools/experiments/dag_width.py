#!/usr/bin/env python3
"""How wide is the headline program's dataflow graph?  Runs the ResNet-20 HEVM program with the plan issued step by step, synchronised
after every step (option step_profile), and prints (stderr of the library) the sum of all step times next to the sum over waves of each
wave's LONGEST step: the second number is what a scheduler with unlimited concurrency inside a wave -- more streams, an explicitly
built HIP graph with the plan's own dependencies -- could reach at best.     python tools/experiments/dag_width.py [fixture=resnet20]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import runner  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "resnet20"
fx = ha.read_fixture(ROOT / "tests" / "golden" / name)
runner.set_option("step_profile", 1)
vm = runner.HEVM(fresh=True, logN=15, num_primes=14, vm_options={"plan_graph": 0, "plan_lanes": 1})
vm.load_mem(fx["cst"], fx["hevm"])
vm.setInput(0, fx["packed"])
runner.set_option("step_profile", 0)
vm.run()                       # warm
runner.set_option("step_profile", 1)
vm.run()
vm.close()
# ... and what the two-stream graph gets out of that width today: the same program replayed as a HIP graph on one stream and on two
import time

runner.set_option("step_profile", 0)
for lanes, graph in ((1, 1), (2, 1), (2, 2), (1, 2)):
    vm = runner.HEVM(fresh=True, logN=15, num_primes=14, vm_options={"plan_graph": graph, "plan_lanes": lanes})
    vm.load_mem(fx["cst"], fx["hevm"])
    vm.setInput(0, fx["packed"])
    vm.run()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        vm.run()
        ts.append((time.perf_counter() - t0) * 1e3)
    what = "captured from the streams (fork / join per wave)" if graph == 1 else "built from the plan's dependencies (plan_graph = 2)"
    print(f"graph replay, {lanes} scratch lane(s), {what}: best {min(ts):.2f} ms, median {sorted(ts)[len(ts) // 2]:.2f} ms", file=sys.stderr)
    vm.close()

# the fork / join threshold of the two-stream capture (option plan_aux_min_cost)
for cost in (1, 3, 5, 8, 11, 16):
    vm = runner.HEVM(fresh=True, logN=15, num_primes=14, vm_options={"plan_aux_min_cost": cost})
    vm.load_mem(fx["cst"], fx["hevm"])
    vm.setInput(0, fx["packed"])
    vm.run()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        vm.run()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"two-stream capture, auxiliary share >= {cost} cost units: best {min(ts):.2f} ms, median {sorted(ts)[len(ts) // 2]:.2f} ms", file=sys.stderr)
    vm.close()
