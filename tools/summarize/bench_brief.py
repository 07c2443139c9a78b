#!/usr/bin/env python3
"""one-line digest of bench.py JSON lines read from stdin (experiments on the GPU box)"""
import json
import sys

for ln in sys.stdin:
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    r, c3 = d.get("roofline") or {}, d.get("cfg3_mul_relin") or {}
    print(f"ms_per_step {d['ms_per_step']:.2f}  ntt/s {d['value']:.0f}  roofline.frac {r.get('frac')}  leg_us {(r.get('launch') or {}).get('avg_us')}  "
          f"cfg3_us {c3.get('us')}  rms {((d.get('decrypted_error') or {}).get('rms_vs_torch'))}")
