#!/usr/bin/env python3
"""Only the headline workload (set-up, then `runs` run() calls of the ResNet-20 HEVM program of bench.py), for rocprofv3 passes whose
counters should describe the timed step and nothing else:  python3 tools/legs/headline_only.py [runs=3] [lowering = b6 | b13] [--streams S] [--opt name=value ...]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import runner  # noqa: E402

sys.argv = runner.apply_cli_options(sys.argv)
streams = 1
if "--streams" in sys.argv:  # S independent images per run() in one VM (the fed regime: bench.py's streams leg)
    k = sys.argv.index("--streams")
    streams = int(sys.argv[k + 1])
    del sys.argv[k : k + 2]
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
fx = ha.read_fixture(ROOT / "tests" / "golden" / "resnet20")
if len(sys.argv) > 2:  # another lowering of the same trace (same constants)
    import gzip

    fx["hevm"] = gzip.open(ROOT / "tests" / "golden" / f"resnet20.{sys.argv[2]}.hevm.gz").read()
hevm = runner.HEVM(fresh=True, logN=15, num_primes=14)
if streams > 1:
    hevm.set_streams(streams)
hevm.load_mem(fx["cst"], fx["hevm"])
for q in range(streams):
    if streams > 1:
        hevm.select_stream(q)
    hevm.setInput(0, fx["packed"])
for _ in range(runs):
    hevm.run()
