"""Static statistics of an HEVM program: what one run() must compute, independent of how a backend executes it.

Walks the bytecode in program order with the level semantics of the reference's interpreter
(/root/reference/lib/Runtime/SEAL_HEVM.cpp:268-334) and prices every instruction with SURVEY.md section 8(d)'s table:

    unit                      NTT-equivalents      algorithmic bytes (P_limb = 8 N)
    negate                    0                    4 l
    addcc                     0                    6 l
    addcp / mulcp             0                    5 l
    modswitch by d            0                    4 (l - d)
    rescale                   2 l                  2 l + 2 (l - 1)
    key-switch hop (rotate)   (l + 1)(l + 2)       2 l^2 + 7 l
    mulcc + relinearise       (l + 1)(l + 2)       4 l + 2 l^2 + 7 l
    opcode 10 (re-encrypt)    0 (not counted)      2 l + 2 t          (read the operand, write the result)

A rotation costs one hop per non-zero NAF digit of its offset unless the default Galois key set
(KeyGenerator::create_galois_keys: steps +-2^k and conjugation, SEAL_HEVM.cpp:82-83) holds a direct key
[SEAL-upstream Evaluator::rotate_internal].  Used by bench.py for `roofline.step` and by the --dry-run path.
"""
from __future__ import annotations

from collections import Counter

from .hevm_asm import (OP_ADDCC, OP_ADDCP, OP_BOOTSTRAP, OP_CONJ, OP_ENCODE, OP_ENCODE_COMPLEX, OP_MODRAISE, OP_MODSWITCH, OP_MULCC, OP_MULCP,
                       OP_NEGATE, OP_RESCALE, OP_ROTATE, OP_SETSCALE, unpack_hevm)


def naf(value: int):
    """non-adjacent form, least significant digit first, as signed powers of two [SEAL-upstream util::naf]"""
    out, sign, v, bit = [], (-1 if value < 0 else 1), abs(value), 0
    while v:
        if v & 1:
            d = 2 - (v & 3)  # +1 or -1
            out.append(sign * d * (1 << bit))
            v -= d
        v >>= 1
        bit += 1
    return out


def rotate_hops(offset: int, slots: int) -> int:
    """key-switch hops of rotate_vector(offset) under the default Galois key set"""
    if offset == 0:
        return 0
    e = offset % slots  # Galois element 3^e; the default set holds 3^(+-2^k), k = 0 .. log2(slots) - 1
    if e == 0:
        return 0
    if e & (e - 1) == 0 or (slots - e) & (slots - e - 1) == 0:
        return 1
    return sum(rotate_hops(d, slots) for d in naf(offset) if abs(d) != slots)


def BOOTSTRAP_GAP(opc: int) -> bool:
    """opcode numbers 11-15 are unassigned (10 = bootstrap, 16-19 = this runtime's extension opcodes)"""
    return OP_BOOTSTRAP < opc < OP_ENCODE_COMPLEX


def walk(hevm: bytes, logN: int = 15, direct_keys: bool = False) -> dict:
    """direct_keys: a Galois key exists for every rotation offset of the program (hevm_add_rotation_keys): one hop per rotation"""
    h = unpack_hevm(hevm)
    N, slots = 1 << logN, 1 << (logN - 1)
    p_limb = 8 * N
    lvl = {i: int(v) for i, v in enumerate(h["arg_level"])}
    ntts = ks = 0
    limbs = 0  # algorithmic bytes in units of P_limb
    ks_hist, rs_hist, boot_hist, op_bytes = Counter(), Counter(), Counter(), Counter()
    for opc, dst, lhs, rhs in h["ops"].tolist():
        if opc in (OP_ENCODE, OP_ENCODE_COMPLEX) or opc > OP_SETSCALE or BOOTSTRAP_GAP(opc):
            continue
        l = lvl[lhs]
        out_l, b = l, 0
        if opc == OP_ROTATE:
            off = rhs - 65536 if rhs >= 32768 else rhs
            hops = (1 if off % slots else 0) if direct_keys else rotate_hops(off, slots)
            ks += hops
            ks_hist[l] += hops
            ntts += hops * (l + 1) * (l + 2)
            b = hops * (2 * l * l + 7 * l)
        elif opc == OP_NEGATE:
            b = 4 * l
        elif opc == OP_RESCALE:
            ntts += 2 * l
            rs_hist[l] += 1
            b = 2 * l + 2 * (l - 1)
            out_l = l - 1
        elif opc == OP_MODSWITCH:
            down = rhs - 65536 if rhs >= 32768 else rhs
            if down <= 0:
                continue
            out_l = l - down
            b = 4 * out_l
        elif opc == OP_ADDCC:
            b = 6 * l
        elif opc in (OP_ADDCP, OP_MULCP):
            b = 5 * l
        elif opc == OP_MULCC:
            ks += 1
            ks_hist[l] += 1
            ntts += (l + 1) * (l + 2)
            b = 4 * l + 2 * l * l + 7 * l
        elif opc == OP_BOOTSTRAP:
            out_l = rhs
            boot_hist[(l, rhs)] += 1
            b = 2 * l + 2 * rhs
        elif opc == OP_CONJ:      # extension: conjugation = one key switch with the key of Galois element 2N - 1
            ks += 1
            ks_hist[l] += 1
            ntts += (l + 1) * (l + 2)
            b = 2 * l * l + 7 * l
        elif opc == OP_MODRAISE:  # extension: 2 inverse NTTs of one limb, 2 * rhs forward NTTs
            out_l = rhs
            ntts += 2 + 2 * rhs
            b = 2 * l + 2 * rhs
        elif opc == OP_SETSCALE:
            b = 0
        limbs += b
        op_bytes[opc] += b * p_limb
        lvl[dst] = out_l
    return {"ntt_equivalents": ntts, "key_switches": ks, "algorithmic_bytes": limbs * p_limb,
            "key_switch_level_histogram": {str(k): v for k, v in sorted(ks_hist.items())},
            "rescale_level_histogram": {str(k): v for k, v in sorted(rs_hist.items())},
            "opcode10_histogram": {f"{a}->{b}": v for (a, b), v in sorted(boot_hist.items())},
            "algorithmic_bytes_by_opcode": {str(k): v for k, v in sorted(op_bytes.items())},
            "num_ops": int(len(h["ops"]))}
