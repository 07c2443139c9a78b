#!/usr/bin/env python3
"""Writes tests/golden/hostile/: small valid and hostile `.cst` / `.hevm` / SEAL-serialization inputs for the library's host parsers
(dacapo_amd/csrc/wire_parse.cpp, seal_serial.cpp), and manifest.json saying what each must do.  tests/test_host_fuzz.py runs every
file through the AddressSanitizer + UBSan build of those parsers (csrc/Makefile target host_asan) and mutates the valid ones.

The reference's loaders trust their files (SEAL_HEVM.cpp:182-234: fread into vectors sized by the file's own counts; :91-180: SEAL's
load()); these are the inputs that trust would trip over: truncation at every field, counts beyond the data, negative and 2^62 counts,
bad magic, foreign versions, unknown / corrupt / truncated compression, and decompression bombs (zlib and Zstandard).

    python tools/fixtures/make_hostile_corpus.py        (deterministic; everything is a few KB except the two bombs, ~200 KB and 16 KB)"""
import json
import struct
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from oracle import seal_format as sf  # noqa: E402

OUT = ROOT / "tests" / "golden" / "hostile"
OUT.mkdir(parents=True, exist_ok=True)
manifest = []


def emit(name, kind, data, expect, message="", constants=None):
    (OUT / name).write_bytes(data)
    manifest.append({"file": name, "kind": kind, "expect": expect, "message": message, **({"constants": constants} if constants else {})})


# ---- .cst : i64 count | count x (i64 length | length x f64) ------------------------------------------------------------------------
good_cst = ha.pack_cst([np.array([0.75]), np.arange(3, dtype=np.float64), np.linspace(-1, 1, 16), np.array([1 << 40], dtype=np.float64)])
emit("cst_valid.cst", "cst", good_cst, "ok")
emit("cst_empty_file.cst", "cst", b"", "rejected", "truncated")
emit("cst_zero_constants.cst", "cst", struct.pack("<q", 0), "ok")
emit("cst_count_only.cst", "cst", struct.pack("<q", 3), "rejected", "implausible constant count")
emit("cst_negative_count.cst", "cst", struct.pack("<q", -1) + good_cst[8:], "rejected", "implausible constant count")
emit("cst_count_2_62.cst", "cst", struct.pack("<q", 1 << 62) + good_cst[8:], "rejected", "implausible constant count")
emit("cst_length_2_62.cst", "cst", good_cst[:8] + struct.pack("<q", 1 << 62) + good_cst[16:], "rejected", "claims")
emit("cst_negative_length.cst", "cst", good_cst[:8] + struct.pack("<q", -5) + good_cst[16:], "rejected", "claims")
emit("cst_length_overflows_by_one.cst", "cst", good_cst[:-8], "rejected", "claims")
emit("cst_truncated_mid_length.cst", "cst", good_cst[:8 + 8 + 8 + 3], "rejected", "")
emit("cst_one_constant_cut_in_its_length.cst", "cst", struct.pack("<q", 1) + b"\x02\x00\x00", "rejected", "")
emit("cst_one_constant_cut_in_its_values.cst", "cst", struct.pack("<qq", 1, 2) + struct.pack("<d", 1.5) + b"\x00\x00", "rejected", "claims")

# ---- .hevm : HEVMHeader | ConfigBody | arg / res tables | ops -----------------------------------------------------------------------
E, ROT, ADDCP, MULCP, ADDCC, SETSCALE = ha.OP_ENCODE, ha.OP_ROTATE, ha.OP_ADDCP, ha.OP_MULCP, 6, 19
ops = [(E, 0, 0, (3 << 10) + 30), (ROT, 1, 0, 5), (MULCP, 1, 1, 0), (ADDCC, 2, 1, 0), (ADDCP, 2, 2, 0)]
good = ha.pack_hevm([30], [3], [30], [3], [2], 3, 1, 3, np.array(ops, dtype=np.uint16))
emit("hevm_valid.hevm", "hevm", good, "ok", constants="cst_valid.cst")
emit("hevm_valid_header_only.hevm", "hevm-header", good, "ok")
hdr = struct.Struct("<IIQQ")     # magic, header size, arg_length, res_length
cfg = struct.Struct("<QQQQQ")    # config_body_length, num_operations, num_ctxt_buffer, num_ptxt_buffer, init_level
h, c = list(hdr.unpack_from(good, 0)), list(cfg.unpack_from(good, hdr.size))
body = good[hdr.size + cfg.size:]


def rebuild(hh=None, cc=None, tail=None):
    return hdr.pack(*(hh or h)) + cfg.pack(*(cc or c)) + (body if tail is None else tail)


emit("hevm_empty_file.hevm", "hevm", b"", "rejected", "truncated")
for cut in (3, hdr.size - 1, hdr.size + 7, hdr.size + cfg.size - 1, hdr.size + cfg.size + 5, len(good) - 1, len(good) - 8):
    emit(f"hevm_truncated_at_{cut}.hevm", "hevm", good[:cut], "rejected", "")
emit("hevm_bad_magic.hevm", "hevm", rebuild(hh=[0x4D564548] + h[1:]), "rejected", "magic")
emit("hevm_args_2_60.hevm", "hevm", rebuild(hh=h[:2] + [1 << 60, h[3]]), "rejected", "more arguments")
emit("hevm_results_2_63.hevm", "hevm", rebuild(hh=h[:3] + [1 << 63]), "rejected", "more arguments")
emit("hevm_ops_2_61.hevm", "hevm", rebuild(cc=[c[0], 1 << 61] + c[2:]), "rejected", "more arguments")
emit("hevm_ops_one_too_many.hevm", "hevm", rebuild(cc=[c[0], c[1] + 1] + c[2:]), "rejected", "truncated")
emit("hevm_70000_cipher_registers.hevm", "hevm", rebuild(cc=c[:2] + [70000] + c[3:]), "rejected", "more arguments")
emit("hevm_2_40_plain_registers.hevm", "hevm", rebuild(cc=c[:3] + [1 << 40] + c[4:]), "rejected", "more arguments")


def with_ops(new_ops, nptxt=1):
    return ha.pack_hevm([30], [3], [30], [3], [2], 3, nptxt, 3, np.array(new_ops, dtype=np.uint16))


emit("hevm_encode_into_missing_plain.hevm", "hevm", with_ops([(E, 7, 0, (3 << 10) + 30)]), "rejected", "encode into plaintext register")
emit("hevm_mulcp_reads_missing_plain.hevm", "hevm", with_ops(ops[:2] + [(MULCP, 1, 1, 9)]), "rejected", "reads plaintext register")
emit("hevm_addcp_reads_missing_plain.hevm", "hevm", with_ops(ops[:2] + [(ADDCP, 1, 1, 65535)]), "rejected", "reads plaintext register")
emit("hevm_zero_plain_registers.hevm", "hevm", with_ops(ops, nptxt=0), "rejected", "plaintext register")
emit("hevm_cipher_register_65535.hevm", "hevm", with_ops([(ROT, 65535, 0, 1), (ADDCC, 65535, 65535, 65534)]), "ok")
emit("hevm_unknown_opcodes.hevm", "hevm", with_ops([(77, 9999, 9999, 9999), (0xFFFF, 0, 0, 0), (15, 1, 2, 3)]), "ok")
emit("hevm_setscale_without_constants.hevm", "hevm", with_ops([(SETSCALE, 1, 0, 0)]), "rejected", "setscale")
emit("hevm_setscale_constant_out_of_range.hevm", "hevm", with_ops([(SETSCALE, 1, 0, 40)]), "rejected", "setscale", constants="cst_valid.cst")
emit("hevm_setscale_ok.hevm", "hevm", with_ops([(SETSCALE, 1, 0, 3)]), "ok", constants="cst_valid.cst")
neg_cst = ha.pack_cst([np.array([-2.0])])
emit("cst_negative_scale.cst", "cst", neg_cst, "ok")
emit("hevm_setscale_negative_scale.hevm", "hevm", with_ops([(SETSCALE, 1, 0, 0)]), "rejected", "setscale", constants="cst_negative_scale.cst")
res_tail = bytearray(body)
struct.pack_into("<Q", res_tail, 16 + 16, 1 << 20)  # res_dst[0] (after arg_scale, arg_level, res_scale, res_level: 4 x 8 bytes)
emit("hevm_result_register_2_20.hevm", "hevm", rebuild(tail=bytes(res_tail)), "rejected", "result register")

# ---- SEAL 4.0 serialization ---------------------------------------------------------------------------------------------------------
N, primes = 16, [(1 << 60) - 93 * 32 + 1, (1 << 60) - 173 * 32 + 1, (1 << 60) - 425 * 32 + 1]  # (shape only: the parsers do no arithmetic)
rng = np.random.default_rng(5)
pid = sf.parms_id(N, primes)
params = sf.params_members(N, primes)
ct = sf.ciphertext_members(pid, rng.integers(0, 1 << 60, (2, 3, N), dtype=np.uint64), scale=2.0**40)
pt = sf.plaintext_members(pid, rng.integers(0, 1 << 60, 3 * N, dtype=np.uint64))
key = rng.integers(0, 1 << 60, (2, 2, 3, N), dtype=np.uint64)  # [digits][2][K][N]
ksk = sf.kswitch_members(pid, 3, {1: key})
emit("seal_params.seal", "seal", sf.wrap(params), "ok", "params")
emit("seal_params_zlib.seal", "seal", sf.wrap(params, sf.COMPR_ZLIB), "ok", "params")
emit("seal_ciphertext.seal", "seal", sf.wrap(ct), "ok", "ciphertext")
emit("seal_ciphertext_zlib.seal", "seal", sf.wrap(ct, sf.COMPR_ZLIB), "ok", "ciphertext")
emit("seal_plaintext.seal", "seal", sf.wrap(pt), "ok", "plaintext")
emit("seal_kswitchkeys.seal", "seal", sf.wrap(ksk), "ok", "kswitchkeys")
emit("seal_kswitchkeys_zlib.seal", "seal", sf.wrap(ksk, sf.COMPR_ZLIB), "ok", "kswitchkeys")
good_seal = sf.wrap(ct)
emit("seal_empty_file.seal", "seal", b"", "rejected", "truncated")
emit("seal_header_cut.seal", "seal", good_seal[:9], "rejected", "truncated")
emit("seal_bad_magic.seal", "seal", b"\xa1\x5e" + good_seal[2:], "rejected", "bad magic")
emit("seal_header_size_32.seal", "seal", good_seal[:2] + b"\x20" + good_seal[3:], "rejected", "bad magic")
emit("seal_version_3_6.seal", "seal", good_seal[:3] + b"\x03\x06" + good_seal[5:], "rejected", "SEAL 3.x")
emit("seal_unknown_compr_mode.seal", "seal", good_seal[:5] + b"\x07" + good_seal[6:], "rejected", "compr_mode")
emit("seal_size_field_beyond_data.seal", "seal", good_seal[:8] + struct.pack("<Q", len(good_seal) + 1) + good_seal[16:], "rejected", "exceeds")
emit("seal_size_field_2_63.seal", "seal", good_seal[:8] + struct.pack("<Q", 1 << 63) + good_seal[16:], "rejected", "exceeds")
emit("seal_size_field_below_header.seal", "seal", good_seal[:8] + struct.pack("<Q", 7) + good_seal[16:], "rejected", "exceeds")
emit("seal_members_cut.seal", "seal", sf.header(16 + len(ct) - 40) + ct[:-40], "ok", "")  # no member parser accepts it: parsed=[]
z = zlib.compress(ct)
emit("seal_zlib_corrupt.seal", "seal", sf.header(16 + len(z), sf.COMPR_ZLIB) + z[:20] + bytes(b ^ 0x5A for b in z[20:60]) + z[60:], "rejected", "zlib")
emit("seal_zlib_truncated.seal", "seal", sf.header(16 + len(z) - 9, sf.COMPR_ZLIB) + z[:-9], "rejected", "zlib")
emit("seal_zlib_garbage.seal", "seal", sf.header(16 + 64, sf.COMPR_ZLIB) + bytes(range(64)), "rejected", "zlib")
# the one legitimate object that deflates ~1000 : 1: an all-zero ciphertext.  At the largest ring the library runs (N = 2^17, here 31 limbs: 62 MiB
# -> ~62 KB) it must load (round-5 advisor: the limit sized for the reference's ring refused it)
big_pid = sf.parms_id(1 << 17, list(range(3, 3 + 31)))
zero_ct = sf.ciphertext_members(big_pid, np.zeros((2, 31, 1 << 17), dtype=np.uint64), scale=2.0**40)
emit("seal_ciphertext_all_zero_N131072_zlib.seal", "seal", sf.header(16 + len(zlib.compress(zero_ct, 9)), sf.COMPR_ZLIB) + zlib.compress(zero_ct, 9), "ok", "ciphertext")
bomb = zlib.compress(bytes(200 << 20), 9)  # 200 MiB of zeros -> ~200 KB: 1030 : 1, far beyond the 64 x + 128 MiB a key object may expand to
emit("seal_zlib_bomb_200MiB.seal", "seal", sf.header(16 + len(bomb), sf.COMPR_ZLIB) + bomb, "rejected", "expand")
# Zstandard, hand-assembled (RFC 8878): magic | frame header descriptor 0x00 (no content size, windowed) | window descriptor (128 KiB) |
# 4 000 RLE blocks of 128 KiB each (3-byte block header: last | type 1 << 1 | size << 3, + the byte) = 16 KB -> 500 MiB
blocks = b"".join(struct.pack("<I", ((1 << 17) << 3) | (1 << 1) | (1 if i == 3999 else 0))[:3] + b"\x00" for i in range(4000))
zbomb = struct.pack("<I", 0xFD2FB528) + b"\x00" + bytes([7 << 3]) + blocks
emit("seal_zstd_bomb_500MiB.seal", "seal", sf.header(16 + len(zbomb), sf.COMPR_ZSTD) + zbomb, "rejected", "")  # "expand", or "libzstd.so.1 is not available"
emit("seal_zstd_garbage.seal", "seal", sf.header(16 + 64, sf.COMPR_ZSTD) + bytes(range(64)), "rejected", "")
# members whose own counts lie
ct_hdr = 32 + 1 + 8 * 4 + 8  # parms_id | is_ntt | size, N, limbs, correction factor | scale
liar = bytearray(ct)
struct.pack_into("<Q", liar, 33, 1 << 40)  # size
emit("seal_ciphertext_size_2_40.seal", "seal", sf.wrap(bytes(liar)), "ok", "")  # rejected as a ciphertext ("implausible"); nothing else parses it either
liar = bytearray(ct)
struct.pack_into("<Q", liar, 33 + 8, 1 << 19)  # N: size * limbs * N no longer the array's length
emit("seal_ciphertext_degree_lies.seal", "seal", sf.wrap(bytes(liar)), "ok", "")
liar = bytearray(ct)
struct.pack_into("<Q", liar, ct_hdr + 16, 1 << 61)  # the nested DynArray's count (after its own 16-byte header)
emit("seal_dynarray_count_2_61.seal", "seal", sf.wrap(bytes(liar)), "ok", "")
liar = bytearray(ksk)
struct.pack_into("<Q", liar, 32, 1 << 30)  # dim1
emit("seal_kswitch_dim1_2_30.seal", "seal", sf.wrap(bytes(liar)), "ok", "")
liar = bytearray(ksk)
struct.pack_into("<Q", liar, 32 + 8 + 8, 1 << 50)  # dim2 of entry 1
emit("seal_kswitch_dim2_2_50.seal", "seal", sf.wrap(bytes(liar)), "ok", "")
liar = bytearray(params)
struct.pack_into("<Q", liar, 1 + 8, 1 << 33)  # coeff_modulus_size
emit("seal_params_2_33_primes.seal", "seal", sf.wrap(bytes(liar)), "ok", "")

(OUT / "manifest.json").write_text(json.dumps({"generator": "tools/fixtures/make_hostile_corpus.py", "files": manifest}, indent=1) + "\n")
print(len(manifest), "files,", sum((OUT / m["file"]).stat().st_size for m in manifest) >> 10, "KiB ->", OUT)
