#!/usr/bin/env python3
"""bench.py's per-op leg (rotation hop, ct x ct + relinearise, rescale at 13 primes, N = 2^15) and config 3 alone, for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/per_op_only.py [iters=20] [--opt name=value ...]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from dacapo_amd import lowlevel as ll  # noqa: E402
from dacapo_amd import runner  # noqa: E402

sys.argv = runner.apply_cli_options(sys.argv)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print(json.dumps({"per_op_13_primes": bench.per_op_leg(ll, iters=iters), "cfg3": bench.cfg3_leg(ll, iters=max(5, iters // 2))}))
