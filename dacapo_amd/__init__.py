"""dacapo_amd -- MI355X-native HEVM runtime: a drop-in for the reference's libSEAL_HEVM.so.

The product is the shared library dacapo_amd/lib/libSEAL_HEVM.so (HIP kernels + C++ host, built from
dacapo_amd/csrc by `__graft_entry__.build()`); this package only holds ctypes bindings:

  * dacapo_amd.runner   -- mirror of the reference's python/hecate/hecate/runner.py (class HEVM)
  * dacapo_amd.lowlevel -- the kernel-level C ABI of include/dacapo_ckks.h
  * dacapo_amd.hevm_asm -- .hevm/.cst writer (the reference's emitter is an MLIR pass we cannot run here)

There is no CPU fallback: importing the bindings without the built library, or creating a context without
a GPU, fails loudly.
"""
import os
from pathlib import Path

_LIB_DIR = Path(__file__).resolve().parent / "lib"
# The RELEASE builds -- what a maintainer deploys, what bench.py and __graft_entry__.smoke() run: the reference's 18 symbols + the safe extensions.
LIB_PATH_RELEASE, LIB_PATH_GW_RELEASE = _LIB_DIR / "libSEAL_HEVM.so", _LIB_DIR / "libSEAL_HEVM_gw.so"
# The HOOKS builds: the same objects + csrc/test_hooks.hip (seeded keys, secret-key pointer, zero encryptions).  tests/conftest.py selects them
# with DACAPO_AMD_HOOKS=1 before this package is imported; nothing else does.
HOOKS = os.environ.get("DACAPO_AMD_HOOKS") == "1"
# DACAPO_AMD_LIB: another build of the same library (kernel-tuning sweeps: tools/experiments/sweep_lds_pad.sh); never a different backend
LIB_PATH = Path(os.environ.get("DACAPO_AMD_LIB") or (_LIB_DIR / "libSEAL_HEVM_hooks.so" if HOOKS else LIB_PATH_RELEASE))
# the generic-width build of the same sources (csrc/modarith.hpp DC_GENERIC_WIDTH = 1): primes of 45..60 bits, mixed chains
LIB_PATH_GW = _LIB_DIR / "libSEAL_HEVM_gw_hooks.so" if HOOKS else LIB_PATH_GW_RELEASE
