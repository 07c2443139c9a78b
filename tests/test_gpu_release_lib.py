"""The RELEASE build of the library (dacapo_amd/lib/libSEAL_HEVM.so: the 18 reference symbols + the safe extensions, no test hooks) is what a
maintainer deploys; the rest of tests/ runs on the hooks build of the same objects (tests/conftest.py).  Here the release build itself goes
through the reference's call sequence in a child process that never sees DACAPO_AMD_HOOKS: __graft_entry__.smoke() -- kernel-level ops and a
whole HEVM program, keys written by the checker as a SEAL-format key directory and loaded with initFullVM (SEAL_HEVM.cpp:404-409), result limbs
== the oracle VM's -- and a VM with fresh keys (hevm_init_fresh) through encrypt -> run -> decrypt_result."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _child(code, timeout=900):
    env = {k: v for k, v in os.environ.items() if k != "DACAPO_AMD_HOOKS"}
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % str(ROOT) + code], env=env, capture_output=True, text=True,
                          timeout=timeout, cwd=str(ROOT))


def test_smoke_runs_on_the_release_build():
    out = _child("import __graft_entry__ as g\ng.smoke()\n")
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "bit-exact vs oracle VM" in out.stdout and "release build" in out.stdout


def test_fresh_keys_and_seed_refusal_on_the_release_build():
    code = '''
import numpy as np
import dacapo_amd as pkg
from dacapo_amd import hevm_asm as ha, runner
assert pkg.LIB_PATH.name == "libSEAL_HEVM.so" and not runner.reinit_lw().has_test_hooks
try:
    runner.HEVM(seed=1, logN=12, num_primes=4)
    print("SEED ACCEPTED")
except RuntimeError as e:
    print("seed refused")
vm = runner.HEVM(fresh=True, logN=12, num_primes=4)
x = np.random.default_rng(1).uniform(-1, 1, vm.slots)
b = ha.Builder(slots=vm.slots, init_level=3)
v = b.input(x)
b.output(b.add_plain(b.mul(b.mul(v, b.rotate(v, 5)), v), [0.125]))
cst, hv, _ = b.assemble()
vm.load_mem(cst, hv)
vm.setInput(0, x)
vm.run()
err = float(np.sqrt(np.mean((vm.getOutput()[0] - b.expected()[0]) ** 2)))
print("rms", err)
assert err < 1e-4, err
vm2 = runner.HEVM(fresh=True, logN=12, num_primes=4)
assert vm.keyDigest() != vm2.keyDigest(), "two fresh VMs must not share keys"
print("ok")
'''
    out = _child(code)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "seed refused" in out.stdout and "SEED ACCEPTED" not in out.stdout and out.stdout.strip().endswith("ok")
