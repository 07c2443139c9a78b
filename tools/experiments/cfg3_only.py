#!/usr/bin/env python3
"""Runs only BASELINE config 3 (one ct x ct multiply + relinearise, N = 2^16, 24 + 1 primes) `iters` times, for rocprofv3:
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/experiments/cfg3_only.py [iters]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from dacapo_amd import lowlevel as ll  # noqa: E402
from dacapo_amd import runner  # noqa: E402

sys.argv = runner.apply_cli_options(sys.argv)  # --opt name=value

print(bench.cfg3_leg(ll, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 5))
