#!/usr/bin/env python3
"""HBM traffic of the NTT phase kernels from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass:
MI355X_MICROARCH.md "rocprofv3 PMC slots") of `python3 tools/legs/ntt_only.py 15 4096 2`:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- python3 tools/legs/ntt_only.py 15 4096 2
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python3 tools/legs/ntt_only.py 15 4096 2
    python tools/summarize/collect_traffic.py out_f/*/*counter_collection.csv out_w/*/*counter_collection.csv > profiles/r02_ntt_hbm_traffic.json
Units and corrections as the guide prescribes: both counters are in KB; on gfx950 FETCH_SIZE counts half of the bytes of a wide
coalesced streaming read (128-B requests tallied at 64 B) and is doubled; WRITE_SIZE is exact."""
import collections
import csv
import hashlib
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent


def per_launch(path, counter):
    acc, calls, seen = collections.defaultdict(float), collections.Counter(), set()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dacapo::", "").replace("dacapo::", "")
            if not (name.startswith("ntt_phase_kernel") or name.startswith("ntt_full15_kernel")):  # nothing else runs on 4096 limbs
                continue
            acc[name] += float(r["Counter_Value"])
            if (name, r["Dispatch_Id"]) not in seen:
                seen.add((name, r["Dispatch_Id"]))
                calls[name] += 1
    return {k: v / calls[k] for k, v in acc.items()}


fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
hbm = {k: (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0 for k in fetch}
# forward transform: the single-crossing kernel ntt_full15_kernel<false, ...> (round 3: what dc_ntt_forward launches on 4096 limbs of N = 2^15),
# else the two phase kernels with INV = false
fwd = [k for k in hbm if k.startswith("ntt_full15_kernel<false")] or [k for k in hbm if re.search(r"<\d, \d, (true|false), false, ", k)]
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python3 tools/legs/ntt_only.py 15 4096 2   (dc_ntt_forward / "
                  "dc_ntt_inverse: forward = the single-crossing kernel, inverse = the two-launch tiles)",
       "lib_sha256": hashlib.sha256((ROOT / "dacapo_amd" / "lib" / "libSEAL_HEVM.so").read_bytes()).hexdigest(),
       "limbs": 4096, "N": 32768,
       "note": "gfx950: FETCH_SIZE counts half of a wide coalesced read (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact; KB -> bytes",
       "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write, "hbm_bytes_per_launch": hbm,
       "forward_ntt_hbm_bytes": sum(hbm[k] for k in fwd), "forward_ntt_kernels": fwd, "forward_ntt_algorithmic_bytes": 2 * 4096 * 32768 * 8}
print(json.dumps(out, indent=1))
