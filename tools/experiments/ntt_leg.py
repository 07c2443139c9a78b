#!/usr/bin/env python3
"""Only bench.py's roofline leg (forward NTT over 4096 limbs of N = 2^15) + config 3: python tools/experiments/ntt_leg.py"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from dacapo_amd import lowlevel as ll  # noqa: E402

r = bench.roofline_leg(ll, ll.Context(15, 14))
c = bench.cfg3_leg(ll)
print(json.dumps({"leg_us": r["launch"]["avg_us"], "frac": r["frac"], "ntt_per_s": r["launch"]["ntt_per_s"], "cfg3_us": c["us"]}))
