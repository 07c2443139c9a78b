#!/usr/bin/env python3
"""The per-op leg (13 primes, N = 2^15) and config 3 under a list of launch-shape option sets, HIP events, one process:
    python tools/legs/per_op_sweep.py [iters=30] "name=value,name=value" "..." ...
Each argument is one option set (applied on top of the defaults, reset afterwards); the first row is always the defaults.
Prints one line per set: rotate hop / ct x ct + relinearise / rescale (us), config 3 (us)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from dacapo_amd import lowlevel as ll  # noqa: E402
from dacapo_amd import runner  # noqa: E402

args = sys.argv[1:]
iters = 30
if args and args[0].isdigit():
    iters = int(args.pop(0))
cfg3 = "--no-cfg3" not in args
args = [a for a in args if a != "--no-cfg3"]
L = ll.lib()
for spec in [""] + args:
    L.hevm_reset_options()
    for kv in filter(None, spec.split(",")):
        k, _, v = kv.partition("=")
        runner.set_option(k, int(v, 0))
    r = bench.per_op_leg(ll, iters=iters)
    c = bench.cfg3_leg(ll, iters=max(5, iters // 3))["us"] if cfg3 else float("nan")
    print(f"{spec or 'defaults':60s} rotate {r['rotate_hop']['us']:7.1f}  mul+relin {r['mulcc_relin']['us']:7.1f}  rescale {r['rescale']['us']:6.1f}  cfg3 {c:7.1f}", flush=True)
L.hevm_reset_options()
