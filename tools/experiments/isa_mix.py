#!/usr/bin/env python3
"""Instruction mix per kernel from a hipcc -S listing (development aid): python tools/experiments/isa_mix.py file.s [filter]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\S+):\s*; @\S+\n(.*?)\n\.Lfunc_end", s, re.S | re.M):
    name, body = m.groups()
    if flt not in name:
        continue
    c = collections.Counter(re.findall(r"^\s+([a-z_0-9]+)", body, re.M))
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    pick = lambda pre: {k: v for k, v in c.items() if k.startswith(pre)}
    print(name[:90])
    print(f"  VALU={valu} mad_u64_u32={c['v_mad_u64_u32']} mul_lo={c['v_mul_lo_u32']} mul_hi={c['v_mul_hi_u32']}"
          f" salu={sum(v for k, v in c.items() if k.startswith('s_'))}")
    print(f"  global={pick('global_')} ds={pick('ds_')} sload={pick('s_load')} barrier={c['s_barrier']}")
