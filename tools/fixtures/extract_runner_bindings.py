#!/usr/bin/env python3
"""Extracts the ctypes bindings the reference's Python driver declares for libSEAL_HEVM.so
(/root/reference/python/hecate/hecate/runner.py:34-71: `lw.<symbol>.argtypes = [...]`, `.restype = ...`) into
tests/golden/runner_bindings.json -- data: symbol names with argument and result types as the CALLER sees them.
tests/test_host_formats.py checks the library and include/hevm_abi.h against it.  Runs in the build container only."""
import json
import re
import sys
from pathlib import Path

src = Path("/root/reference/python/hecate/hecate/runner.py").read_text()
out = {}
for m in re.finditer(r"^lw\.(\w+)\.argtypes\s*=\s*\[(.*?)\]", src, re.M):
    args = [a.strip().replace("ctypes.", "") for a in re.split(r",\s*(?![^()]*\))", m.group(2)) if a.strip()]
    out.setdefault(m.group(1), {})["argtypes"] = args
for m in re.finditer(r"^lw\.(\w+)\.restype\s*=\s*([\w.]+)", src, re.M):
    out.setdefault(m.group(1), {})["restype"] = m.group(2).replace("ctypes.", "")
for v in out.values():
    v.setdefault("restype", "c_int")  # ctypes' default when the driver sets none (the C functions return void)
dst = Path(__file__).resolve().parents[2] / "tests" / "golden" / "runner_bindings.json"
dst.write_text(json.dumps({"source": "python/hecate/hecate/runner.py:34-71", "bindings": out}, indent=1))
print(len(out), "symbols ->", dst)
