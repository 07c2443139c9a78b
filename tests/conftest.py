import os
import sys
from pathlib import Path

import pytest

# tests/ run on the HOOKS builds of the library (libSEAL_HEVM_hooks.so, libSEAL_HEVM_gw_hooks.so: the release builds' objects + csrc/test_hooks.hip --
# seeded keys, the secret-key pointer, zero encryptions).  Must be set before dacapo_amd is imported; child processes inherit it.  The release
# builds' own checks -- export lists, and the reference's call sequence through the library a maintainer deploys -- name them explicitly
# (tests/test_host_formats.py, tests/test_gpu_release_lib.py: a child process with DACAPO_AMD_HOOKS unset).
os.environ.setdefault("DACAPO_AMD_HOOKS", "1")

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="module", autouse=True)
def _release_gpu_vms_after_each_module():
    """The reference never frees a VM (its ABI has no destroy symbol) and neither do most tests; one module's VMs (tens of GB of direct
    Galois keys in the bootstrapping tests) must not crowd the next module's out of HBM: hevm_destroy whatever is still alive."""
    yield
    mod = sys.modules.get("dacapo_amd.runner")
    if mod is not None:
        mod.close_all()


@pytest.fixture(scope="session")
def oracle_small():
    """Small ring (N=2^10, 4 x 60-bit SEAL-style primes): O(N^2) definitions stay cheap."""
    from oracle.oracle import Oracle

    return Oracle(10, 4)


@pytest.fixture(scope="session")
def oracle_mid():
    """N=2^12, 5 primes, with keys: homomorphic identities in seconds."""
    from oracle.oracle import Oracle

    o = Oracle(12, 5)
    o.keygen(seed=0x4845564D)
    return o


@pytest.fixture(scope="session")
def oracle_ref():
    """The reference's parameters (N=2^15, 14 x 60-bit primes, SEAL_HEVM.cpp:39-53) with keys: ~25 s of keygen, shared."""
    from oracle.oracle import Oracle

    o = Oracle(15, 14)
    o.keygen(seed=7)
    return o
