#!/usr/bin/env python3
"""tests/golden/perf_guard.json from a full bench record (bench.py --out / profiles/<round>_bench_full.json): the figures
tests/test_gpu_perf_guard.py holds later builds to, with the box's copy rate beside them.
    python tools/summarize/perf_guard.py profiles/r06_bench_full.json [round]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
rec = json.loads(Path(sys.argv[1]).read_text())
out = {"round": sys.argv[2] if len(sys.argv) > 2 else Path(sys.argv[1]).name.split("_")[0], "source": sys.argv[1],
       "copy_kernel_gbs": rec["roofline"]["copy_kernel_gbs"], "hop13_us": rec["per_op_13_primes"]["rotate_hop"]["us"],
       "mulrelin13_us": rec["per_op_13_primes"]["mulcc_relin"]["us"], "cfg3_us": rec["cfg3_mul_relin"]["us"], "headline_ms": rec["ms_per_step"]}
(ROOT / "tests" / "golden" / "perf_guard.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out))
