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

# DACAPO_AMD_LIB: another build of the same library (kernel-tuning sweeps: tools/experiments/sweep_lds_pad.sh); never a different backend
LIB_PATH = Path(os.environ.get("DACAPO_AMD_LIB") or Path(__file__).resolve().parent / "lib" / "libSEAL_HEVM.so")
# the generic-width build of the same sources (csrc/modarith.hpp DC_GENERIC_WIDTH = 1): primes of 45..60 bits, mixed chains
LIB_PATH_GW = Path(__file__).resolve().parent / "lib" / "libSEAL_HEVM_gw.so"
