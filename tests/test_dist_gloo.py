"""CPU, world_size 2, gloo: the multi-GPU path of bench.py (stream sharding + whole-job aggregation)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import sys, json
sys.path.insert(0, %r)
from dacapo_amd.dist import Group, streams_of_rank
g = Group(backend="gloo")
mine = streams_of_rank(5, g.rank, g.world)
g.barrier()
# rank r "ran" len(mine) streams of 1000 work units each in (1 + r) seconds
elapsed, work = g.job_totals(1.0 + g.rank, 1000.0 * len(mine))
print(json.dumps({"rank": g.rank, "mine": mine, "elapsed": elapsed, "work": work}), flush=True)
g.close()
"""


def test_two_rank_gloo_sharding_and_aggregation(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % str(ROOT))
    import socket
    import time

    for attempt in range(3):  # the rendezvous port is picked free but can be taken in between: retry with another one
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                              "127.0.0.1", "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=600)
        if out.returncode == 0:
            break
        time.sleep(2.0)
    assert out.returncode == 0, out.stderr[-2000:]
    import json

    import re

    # both ranks write to the same pipe: their lines can land on one line, so pick the objects out instead of splitting lines
    rows = [json.loads(m) for m in re.findall(r"\{[^{}]*\}", out.stdout)]
    assert len(rows) == 2
    by_rank = {r["rank"]: r for r in rows}
    assert by_rank[0]["mine"] == [0, 2, 4] and by_rank[1]["mine"] == [1, 3]  # stream s -> rank s mod world
    for r in rows:  # every rank sees the job totals: max time, summed work
        assert r["elapsed"] == 2.0 and r["work"] == 5000.0


def test_single_process_group_is_a_noop():
    from dacapo_amd.dist import Group

    g = Group()
    assert g.world == 1 and g.job_totals(0.5, 7.0) == (0.5, 7.0)
    g.barrier()
    g.close()


def _bench_lines(args, env=None):
    import json

    out = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts two ranks as a child torch.distributed.run and
    exactly one JSON line (rank 0's) reports the whole job.  --dry-run replaces the device work by a sleep and RCCL by gloo;
    argument parsing, rank discovery, barrier, max-over-ranks time and summed work are the real run's code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    rows = _bench_lines(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"], env)
    assert len(rows) == 1
    r = rows[0]
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "weak" and r["dry_run"] is True
    # whole-job value: the work of both ranks over the max-over-ranks time
    per_step = r["config"]["ntt_equivalents_per_step"]
    assert abs(r["value"] - 2 * per_step * r["steps"] / (r["ms_per_step"] * r["steps"] * 1e-3)) / r["value"] < 1e-3


def test_replicas_hold_one_key_set_by_broadcast():
    """keys are replicated over the GPUs.  The release library has no seeded key generation (round 6), so with more than one
    rank the ranks always start from DIFFERENT sets, rank 0's is broadcast buffer by buffer, the digests then agree (the run would abort
    otherwise) and the line reports the bytes shipped.  --streams S rides along on the multi-rank path (config 5 = 8 GPUs x S)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    two = _bench_lines(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run", "--streams", "3"], env)[0]
    assert two["keys"]["keys"] == "shared" and two["keys"]["mode"] == "broadcast" and two["keys"]["broadcast_bytes"] == 3 * 4096 * 8
    assert two["config"]["streams_per_gpu"] == 3
    single = _bench_lines(["--dry-run"], env)[0]
    assert single["keys"]["mode"] == "local" and single["keys"]["broadcast_bytes"] == 0
    assert single["keys"]["digest"] == two["keys"]["digest"]              # rank 0's set is the one every replica ends up with
    bc = _bench_lines(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run", "--broadcast-keys"], env)[0]
    assert bc["keys"] == two["keys"]                                      # the flag only forces the path at world size 1


def test_share_keys_refuses_replicas_with_different_keys(tmp_path):
    worker = tmp_path / "w.py"
    worker.write_text('''
import sys
sys.path.insert(0, %r)
from dacapo_amd.dist import Group
g = Group(backend="gloo")
try:
    g.share_keys(lambda: 1000 + g.rank)          # per-rank digests: must be refused
    print("NOT REFUSED", flush=True)
except RuntimeError as e:
    print("refused:", e, flush=True)
g.close()
''' % str(ROOT))
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(worker)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("refused:") == 2 and "NOT REFUSED" not in out.stdout


def test_bench_under_an_external_launcher_uses_its_world_size():
    """the driver's own form: torch.distributed.run ... bench.py --gpus N (RANK set => no second spawn)"""
    import json
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(rows) == 1 and rows[0]["n_gpus"] == 2


def test_bench_single_rank_dry_run():
    rows = _bench_lines(["--dry-run"])
    assert len(rows) == 1 and rows[0]["n_gpus"] == 1


def test_bench_config5_code_path_two_ranks():
    """BASELINE config 5 = one config-4 stream per GPU: `python bench.py --gpus N --program config4`.  Two gloo ranks through bench.py's own
    launch code with --dry-run: each rank lowers the nt = 2^16 program to real bootstrapping and counts its NTT-equivalents as the real run
    does, the replicas compare key digests, rank 0 reports the whole job (no hardware curve is claimed: measured_on_hardware is false)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    rows = _bench_lines(["--gpus", "2", "--program", "config4", "--steps", "2", "--warmup", "0", "--dry-run"], env)
    assert len(rows) == 1
    r = rows[0]
    assert r["n_gpus"] == 2 and r["dry_run"] and r["measured_on_hardware"] is False and r["scaling"] == "weak"
    c = r["config"]
    assert "config 5" in c["workload"] and c["rotation_keys"] == 286 and c["key_switches_per_step"] == 10631
    assert 1.8e6 < c["ntt_equivalents_per_step"] < 2.1e6
    assert c["keys"]["keys"] == "shared" and c["parallelism"].startswith("replicas x2")
    # whole-job value = both ranks' work over the slower rank's time
    assert abs(r["value"] - 2 * c["ntt_equivalents_per_step"] * r["steps"] / (r["ms_per_step"] * 1e-3 * r["steps"])) < 1e-3 * r["value"]
