#!/usr/bin/env python3
"""bench.py's per-op leg (rotation hop, ct x ct + relinearise, rescale at 13 primes, N = 2^15) and config 3 alone, for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/legs/per_op_only.py [iters=20] [--only rotate_hop|mulcc_relin|rescale|cfg3] [--opt name=value ...]
--only: one op per process, so that a kernel shared by two ops (f_ks_lift_fcols, f_dr_lift_fcols ...) is attributed to the op it ran for."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from dacapo_amd import lowlevel as ll  # noqa: E402
from dacapo_amd import runner  # noqa: E402

sys.argv = runner.apply_cli_options(sys.argv)
only = None
if "--only" in sys.argv:
    i = sys.argv.index("--only")
    only = sys.argv[i + 1]
    del sys.argv[i:i + 2]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out = {}
if only != "cfg3":
    out["per_op_13_primes"] = bench.per_op_leg(ll, iters=iters, only=only)
if only in (None, "cfg3"):
    out["cfg3"] = bench.cfg3_leg(ll, iters=max(5, iters // 2), grouped=only is None)
print(json.dumps(out))
