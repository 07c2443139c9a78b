#!/usr/bin/env python3
"""Register counts per kernel from a `hipcc -S --cuda-device-only` listing (development aid):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only dacapo_amd/csrc/fused_ks.hip -o /tmp/fused_ks.s
    python tools/experiments/kernel_regs.py /tmp/fused_ks.s [regex on the demangled name]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
recs = re.findall(r"- \.agpr_count:\s+(\d+)(?:.*\n)*?\s+\.name:\s+(\S+)(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", s)
names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in recs), capture_output=True, text=True).stdout.splitlines()
for (ag, _, sg, vg, sp), d in zip(recs, names):
    d = re.sub(r"\(.*", "", d).replace("void dacapo::", "")
    if re.search(flt, d):
        print(f"{d:70s} vgpr {vg:>4s} agpr {ag:>3s} sgpr {sg:>3s} spill {sp}")
