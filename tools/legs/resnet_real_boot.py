#!/usr/bin/env python3
"""The reference's ResNet-20 with REAL CKKS bootstrapping at every bootstrap site (the headline program tests/golden/resnet20.*, every
opcode 10 rewritten by dacapo_amd/ckks_boot.lower_bootstraps): BASELINE config 4 in spirit -- the reference
runs it on HEaaN (HEAAN_HEVM.cpp:386-399) at N = 2^17; here on SEAL-style 60-bit primes, N = 2^15, 20 primes, sparse secret.
    python tools/legs/resnet_real_boot.py [direct_keys=1|2|49|<n>] [fixture=resnet20] [logN=15] [msg_bits=4] [lowering=""] [ks_special=1] [ks_alpha=ks_special] [chain=60|mixed|mixed_app] [streams=1]
chain mixed: a HEaaN-style chain -- 60-bit base prime, 51-bit rescale primes, 60-bit special primes (HEAAN_HEVM.cpp:55-56, profiled_HEAAN_GPU.json:
rescalingFactor 51) -- on the generic-width build; the lowering must then have been traced with --rescale-bits 51.
lowering: another lowering of the same trace (tests/golden/<fixture>.<lowering>.hevm.gz, same constants), e.g. b14 = bootstraps placed at the
model script's own hints, restoring 14 primes (38 bootstraps instead of 541); ks_special > 1: grouped-digit hybrid key switching with that
many special primes (dacapo_amd/csrc/hybrid_ks.hip), which is what makes a 31-level chain affordable at N = 2^17.
With fixture resnet20_nt16 (the same model traced at the reference script's own nt = 2^16 slots, examples/benchmarks/ResNet.py:50) and
logN 17 this is the HEaaN runtime's ring (HEAAN_HEVM.cpp:55-56): ~100 GB of one-prime-per-digit Galois keys, sized for one MI355X."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from dacapo_amd import hevm_asm as ha  # noqa: E402
from dacapo_amd import ckks_boot as cb  # noqa: E402
from dacapo_amd import runner  # noqa: E402

# --also-opt name=value[,name=value...] (repeatable): after the run, destroy the VM and run the same lowered program once more in another VM
# created with those VM options set (bench.py: hyb_lazy_sum off / on / on with hyb_double_hoist in ONE process -- the program is lowered and
# the fixture read once)
also = []
while "--also-opt" in sys.argv:
    i = sys.argv.index("--also-opt")
    also.append({kv.partition("=")[0]: int(kv.partition("=")[2], 0) for kv in sys.argv[i + 1].split(",")})
    del sys.argv[i:i + 2]
sys.argv = runner.apply_cli_options(sys.argv)  # --opt name=value (csrc/options.hpp)

direct = int(sys.argv[1]) if len(sys.argv) > 1 else 1
name = sys.argv[2] if len(sys.argv) > 2 else "resnet20"
logN = int(sys.argv[3]) if len(sys.argv) > 3 else 15
msg_bits = int(sys.argv[4]) if len(sys.argv) > 4 else 4
lowering = sys.argv[5] if len(sys.argv) > 5 else ""
ks = int(sys.argv[6]) if len(sys.argv) > 6 else 1
alpha = int(sys.argv[7]) if len(sys.argv) > 7 else ks  # primes per digit (alpha < ks: P exceeds a digit's modulus, the switching noise shrinks by the ratio)
chain = sys.argv[8] if len(sys.argv) > 8 else "60"
streams = int(sys.argv[9]) if len(sys.argv) > 9 else 1  # independent images through one plan (every batched step carries all streams' items)
fx = ha.read_fixture(ROOT / "tests" / "golden" / name)
if lowering:
    import gzip

    fx["hevm"] = gzip.open(ROOT / "tests" / "golden" / f"{name}.{lowering}.hevm.gz").read()
boot_target = {int(r) for o, _, _, r in ha.unpack_hevm(fx["hevm"])["ops"].tolist() if o == ha.OP_BOOTSTRAP}
assert len(boot_target) == 1, "every opcode 10 of the program must restore the same number of primes"
boot_target = boot_target.pop()
KB = boot_target + cb.boot_levels() + ks  # primes left after a bootstrap + the bootstrap's own levels + the special primes
assert fx["meta"]["slots"] == 1 << (logN - 1), "the fixture was traced for another slot count"
# mixed: every rescale prime 51 bits (the HEaaN configuration's literal shape: bootstrapping then runs at a 2^51 scale and keeps ~7 bits
# less); mixed_app: 51-bit primes for the PROGRAM's levels only, the bootstrap's own 17 levels stay 60-bit (its precision is the 60-bit chain's)
if chain == "mixed":
    primes = cb.mixed_prime_chain(logN, [60] + [51] * (KB - ks - 1) + [60] * ks)
elif chain == "mixed_app":
    primes = cb.mixed_prime_chain(logN, [60] + [51] * (boot_target - 1) + [60] * (KB - boot_target))
else:
    primes = None
t0 = time.time()
fx["hevm"], fx["cst"] = cb.lower_bootstraps(fx["hevm"], fx["cst"], logN, KB, msg_bits=msg_bits, ks=ks, primes=primes)
print(f"opcode 10 -> real bootstrapping: {time.time()-t0:.1f} s", flush=True)
h = ha.unpack_hevm(fx["hevm"])
ops = h["ops"]
print(f"{len(ops)} instructions, {h['num_ptxt']} plaintext registers, {int((ops[:, 0] == ha.OP_MODRAISE).sum())} real bootstraps", flush=True)
t0 = time.time()
# direct_keys >= 49: a BOUNDED key set -- the reference's HEaaN runtime serves every rotation of a program from its 49 left-rotation keys
# (HEAAN_HEVM.cpp:58-64,124-126).  49 = exactly that list (31 of its offsets are the +-2^k this VM has anyway); n > 49: that list plus the
# program's most frequently used other offsets up to n keys.  Rotations without a direct key are composed from the set (option rot_compose).
HEAAN_OFFSETS = [1, 2, 3, 4, 5, 6, 7, 8, 16, 24, 32, 64, 96, 128, 160, 192, 224, 256, 512, 768, 1024, 2048, 3072, 4096, 5120, 6144, 7168, 8192,
                 16384, 24576, 32768, 40960, 49152, 57344, 61440, 63488, 64512, 64768, 65024, 65280, 65408, 65472, 65504, 65512, 65520, 65528,
                 65532, 65534, 65535]
bounded = direct >= 49


def one_run(extra_vm_options):
    """one VM: keys, load + preprocess, two runs (the second timed), the decrypted logits; the VM is destroyed before returning"""
    t0 = time.time()
    hevm = runner.HEVM(fresh=True, logN=logN, num_primes=KB, ks_special=ks, ks_alpha=alpha, vm_options=dict({"secret_hw": 64, "rot_compose": int(bounded)}, **extra_vm_options),
                       primes=primes)
    n_keys = 0
    if bounded:
        import collections

        slots = 1 << (logN - 1)
        assert slots == 65536, "the reference's offset list is for 2^16 slots"
        norm = lambda o: (o % slots) - (slots if (o % slots) > slots // 2 else 0)
        offs = sorted({norm(o) for o in HEAAN_OFFSETS})
        used = collections.Counter(norm(int(q)) for o, _, _, q in ops.tolist() if o == ha.OP_ROTATE)
        for off, _ in used.most_common():
            if len(offs) >= direct:
                break
            if off != 0 and off not in offs:
                offs.append(off)
        hevm.addRotationKeys(offs)
        n_keys = len(offs)
        print(f"bounded key set: {n_keys} rotation keys (the reference HEaaN runtime's 49 offsets{' + the most used others' if direct > 49 else ''}), other offsets composed", flush=True)
    elif direct:
        offs = cb.rotation_offsets(fx["hevm"])
        if direct == 2:  # direct keys for the bootstraps' own rotations only (they run at up to 19 primes); the model's rotations run at 1-3 primes,
            offs = cb.rotation_offsets(cb.single_bootstrap_program(logN, target=boot_target, ks=ks)[2])  # where a NAF hop pair under the default keys costs little
        hevm.addRotationKeys(offs)
        n_keys = len(offs)
        print(f"{len(offs)} direct rotation keys", flush=True)
    print(f"context + keys {time.time()-t0:.1f} s", flush=True)
    t0 = time.time()
    if streams > 1:
        hevm.set_streams(streams)
    hevm.load_mem(fx["cst"], fx["hevm"])
    print(f"load + preprocess (encode, plan, graph) {time.time()-t0:.1f} s", flush=True)
    for sidx in range(streams):  # stream s > 0 gets the image scaled by 1 - s / 8: another input, a known expectation up to the activations
        if streams > 1:
            hevm.select_stream(sidx)
        hevm.setInput(0, fx["packed"] if sidx == 0 else fx["packed"] * (1.0 - sidx / 8.0))
    if streams > 1:
        hevm.select_stream(0)
    t0 = time.perf_counter()
    hevm.run()
    dt_first = time.perf_counter() - t0          # includes first-use costs (code objects, the plan's first issue)
    t0 = time.perf_counter()
    hevm.run()                                   # the same program on the same input again (fresh encryption randomness in opcode 10 only)
    dt = time.perf_counter() - t0
    out = hevm.getOutput()[0]
    st = hevm.stats()
    key_limbs = hevm.key_digits * 2 * KB
    res = {"chain": "60-bit" if primes is None else ("mixed: 60-bit base and special primes, 51-bit rescale primes" if chain == "mixed" else
                                                    "mixed_app: 51-bit rescale primes for the program's levels, 60-bit base, bootstrapping and special primes"),
           "prime_bits": [int(q).bit_length() for q in primes] if primes else None,
           "log2_QP": sum(int(q).bit_length() for q in (primes or cb.seal_prime_chain(logN, KB))),
           "rotation_keys": n_keys, "rotation_key_bytes": n_keys * key_limbs * (8 << logN), "rot_compose": bool(bounded),
           "fixture": name + ("." + lowering if lowering else ""), "special_primes": ks, "primes_per_digit": alpha, "bootstrap_restores_primes": boot_target, "N": 1 << logN, "slots": 1 << (logN - 1), "primes": KB, "msg_bits": msg_bits, "instructions": int(len(ops)), "run_s": round(dt, 3), "first_run_s": round(dt_first, 3), "key_switches": st["keyswitches"], "ntt_equivalents": st["ntts"], "ntt_per_s": round(st["ntts"] / dt),
           "real_bootstraps": int((ops[:, 0] == ha.OP_MODRAISE).sum()),
           "rms_vs_torch": float(np.sqrt(np.mean((out[:10] * 32 - fx["torch_result"]) ** 2))),
           "rms_vs_plaintext_evaluation": float(np.sqrt(np.mean((out - fx["expected"]) ** 2))),
           "logits": [round(float(v), 4) for v in out[:10] * 32], "torch": [round(float(v), 4) for v in fx["torch_result"]]}
    if streams > 1:  # throughput mode: the other streams' logits are finite and differ from stream 0's (another image); images per second
        others = []
        for sidx in range(1, streams):
            hevm.select_stream(sidx)
            o = hevm.getOutput()[0][:10] * 32
            others.append([round(float(v), 4) for v in o])
        hevm.select_stream(0)
        res.update(streams=streams, images_per_s=round(streams / dt, 3), s_per_image=round(dt / streams, 3), other_streams_logits=others)
    groups = hevm.lazy_groups()  # option hyb_lazy_sum (--opt hyb_lazy_sum=1): sums of direct-key rotations with one division by P
    res["lazy_sums"] = {"groups": len(groups), "rotations": sum(len(g) for g in groups)}
    hevm.close()  # hevm_destroy: the next VM gets this one's HBM back
    return res


res = one_run({})
if also:
    res["also"] = [dict(one_run(opts), vm_options=opts) for opts in also]
print(json.dumps(res))
