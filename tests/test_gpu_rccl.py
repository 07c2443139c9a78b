"""The RCCL path of bench.py under the driver's eyes, on the one GPU a test box has (the reference's only communication hook
is getCtxt / getResIdx, SEAL_HEVM.cpp:463-473 -- the batch split itself has no collective in the op path).

tests/test_dist_gloo.py drives the launch / rank / aggregation code with gloo and a sleep for the step; what it cannot show is
`init_process_group("nccl")` (RCCL on ROCm), the flat broadcast of the REAL key buffers out of HBM, the digest all-gather on device
tensors and the max / sum all-reduces of the timing.  DACAPO_FORCE_DIST=1 builds the process group at world size 1 too, so all of that
runs here: bench.py is started as a CHILD process (a process that has initialised HIP must not exec another program on this pool; a child
is fine) under the environment an external launcher would give rank 0."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _run_bench(extra, timeout):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, DACAPO_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.pop("DACAPO_AMD_HOOKS", None)  # bench.py runs on the RELEASE build of the library, like the driver's own run of it
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        full = Path(d) / "bench_full.json"
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--broadcast-keys", "--no-cpu-baseline",
               "--no-config4", "--out", str(full)] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and len(lines[0]) <= 4096, r.stdout[-1000:]  # the line the driver parses
        line = json.loads(lines[0])
        if full.exists():  # the headline program: the compact line + the full record (keys, decrypted error, legs)
            rec = json.loads(full.read_text())
            assert rec["value"] == line["value"] and line["roofline"]["frac"] == rec["roofline"]["frac"]
            return rec, r.stderr
        return line, r.stderr  # --program config4 prints its (small) record itself


def test_bench_headline_through_rccl_at_world_size_1():
    """init_process_group("nccl") -> per-rank key set -> one flat dist.broadcast per key buffer (secret, public, relinearisation, 28 Galois
    keys: 2.8 GB out of HBM) -> digest all-gather -> timed run between barriers -> all-reduced (max time, summed work) -> ONE JSON line"""
    line, _ = _run_bench([], 900)
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["scaling"] == "weak" and line["unit"] == "NTT/s"
    keys = line["config"]["keys"]
    assert keys["keys"] == "shared" and keys["mode"] == "broadcast"
    # secret [K][N] + public [2][K][N] + relin and 28 distinct default Galois keys [K-1][2][K][N] each, 8 bytes a word
    K, N = 14, 1 << 15
    assert keys["broadcast_bytes"] == 8 * (K * N + 2 * K * N + 29 * (K - 1) * 2 * K * N)
    assert line["value"] > 1e5 and line["ms_per_step"] > 0 and line["decrypted_error"]["rms_vs_torch"] < 2e-3
    assert line["roofline"]["frac"] > 0.05 and line["config"]["parallelism"].startswith("replicas x1")
    assert len(keys["digest"]) == 16 and int(keys["digest"], 16) != 0  # (fresh keys from the OS's randomness: nothing to recompute it from)


def test_bench_config5_code_path_through_rccl_at_world_size_1():
    """--program config4 (with --gpus 8: BASELINE config 5) on the same path: the N = 2^17 VM's secret, public, relinearisation and 286 + 34
    grouped-digit rotation keys go through the broadcast (~120 GB of buffers, one at a time), digests compared, one real run of the
    38-bootstrap program between the barriers"""
    line, _ = _run_bench(["--program", "config4"], 1500)
    assert line["n_gpus"] == 1 and line["measured_on_hardware"] is True and line["dry_run"] is False
    keys = line["config"]["keys"]
    assert keys["mode"] == "broadcast" and keys["broadcast_bytes"] > 90e9 and len(keys["digest"]) == 16
    assert line["config"]["key_switches_per_step"] > 10000 and line["config"]["rotation_keys"] >= 286
    assert line["decrypted_error"]["rms_vs_torch"] < 2e-3
    assert 1.0 < line["hevm_wall_s"] < 10.0
