#!/usr/bin/env python3
"""Extracts the HEVM instruction encoding the reference's emitter uses (/root/reference/include/hecate/Dialect/CKKS/IR/
CKKSOps.td:60-222: per CKKS op the `op.opcode = N; op.dst = ...; op.lhs = ...; op.rhs = ...;` it writes) into
tests/golden/opcode_table.json -- data: mnemonic, opcode number and the kind of each operand field (cipher register,
plain register, constant index, immediate).  tests/test_host_formats.py checks dacapo_amd.hevm_asm's opcode constants and
operand packing against it.  Runs in the build container only."""
import json
import re
from pathlib import Path

src = re.sub(r"/\*.*?\*/", "", Path("/root/reference/include/hecate/Dialect/CKKS/IR/CKKSOps.td").read_text(), flags=re.S)
table = {}
for m in re.finditer(r'def (\w+)\s*:\s*CKKS_Op<"(\w+)".*?op\.opcode = (\d+);\s*op\.dst = (.*?);\s*op\.lhs = (.*?);.*?op\.rhs = ([^;]*);', src, re.S):
    _, mnem, opc, dst, lhs, rhs = m.groups()

    def kind(e):
        e = e.strip()
        if e.startswith("cipherMap"):
            return "cipher"
        if e.startswith("plainMap"):
            return "plain"
        if e == "0":
            return "zero"
        return "imm:" + re.sub(r"\s+", "", e)
    table[mnem] = {"opcode": int(opc), "dst": kind(dst), "lhs": kind(lhs), "rhs": kind(rhs)}
dst = Path(__file__).resolve().parents[2] / "tests" / "golden" / "opcode_table.json"
dst.write_text(json.dumps({"source": "include/hecate/Dialect/CKKS/IR/CKKSOps.td:60-222", "ops": table}, indent=1))
print(json.dumps(table, indent=1))
