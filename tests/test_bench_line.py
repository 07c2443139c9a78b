"""bench.py's last stdout line is what the driver parses: it must stay a small, flat JSON object (round 5's line had grown to 22.5 KB of
nested legs and prose and the driver recorded `parsed: null`).  The full record of round 5 (profiles/r05_bench.json: every leg at its real
size, prose included) is the stub; the compact line built from it must round-trip through json, carry the contract's fields with `roofline`
and `cpu_baseline`, hold no prose, and fit LINE_LIMIT with room to spare."""
import io
import json
import sys
from contextlib import redirect_stdout
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def full_size_stub():
    full = json.loads((ROOT / "profiles" / "r05_bench.json").read_text())
    # what this round's record adds on top of round 5's: the per-op traffic ratios and the two config-4 figures side by side
    full["ks_traffic"] = {"hop13": {"hbm_bytes": 256.5e6, "algorithmic_bytes": 112.5e6, "traffic_over_algorithmic": 2.28, "source": "profiles/r06_per_op_budget_rotate_hop.json"},
                          "cfg3": {"hbm_bytes": 1596e6, "algorithmic_bytes": 742e6, "traffic_over_algorithmic": 2.15, "source": "profiles/r06_per_op_budget_cfg3.json"}}
    c4 = full["config4_resnet20_nt65536_N131072"]
    c4["bootstraps_reference_plan"] = 19
    c4["lazy_sums"] = dict(c4["lazy_sums_ab"]["on"], run_s=2.782)
    return full


def test_compact_line_of_a_full_size_record_fits_and_round_trips():
    full = full_size_stub()
    assert len(json.dumps(full)) > 5 * bench.LINE_LIMIT, "the stub is meant to be a full-size record"
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT and len(text) < 3000, len(text)
    back = json.loads(text)
    assert back == line
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    assert set(back["config"]) == {"workload", "key_switches_per_step", "ntt_equivalents_per_step", "streams_per_gpu", "parallelism"}
    r = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "step_frac", "dominant_kernel", "dominant_frac"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    c = back["cpu_baseline"]
    assert set(c) == {"value", "unit", "cores", "kind", "sample", "seconds"} and c["kind"] == "port" and c["cores"] == 1
    assert back["config4"]["run_s"] and back["config4"]["lazy_sums_run_s"] and back["config4"]["bootstraps_reference_plan"] == 19
    assert back["ks_traffic_over_algorithmic"] == {"hop13": 2.28, "cfg3": 2.15}
    assert back["cfg3_us"] == full["cfg3_mul_relin"]["us"]

    def strings(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, str):
            yield x

    assert all(len(s) <= 300 for s in strings(back)), "no prose in the line"
    for banned in ("history", "note", "what", "lowerings", "streams", "real_bootstrap", "key_shapes", "chains", "key_sets"):
        assert banned not in back and banned not in back["config"] and banned not in back["roofline"], banned


def test_oversized_fields_are_dropped_rather_than_emitted():
    full = full_size_stub()
    full["config"]["workload"] = "w" * 5000          # truncated to 300
    full["roofline"]["step"]["dominant"]["kernel"] = "k" * 9000  # truncated to 80
    full["cpu_baseline"]["sample"] = "s" * 9000      # truncated to 200
    line = bench.compact_line(full)
    assert len(json.dumps(line)) <= bench.LINE_LIMIT
    assert "roofline" in line and "cpu_baseline" in line


def test_emit_prints_the_compact_line_last_and_writes_the_full_record(tmp_path):
    full = full_size_stub()
    buf = io.StringIO()
    with redirect_stdout(buf):
        path = bench.emit(full, tmp_path / "bench_full.json")
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= bench.LINE_LIMIT
    line = json.loads(lines[0])
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    rec = json.loads(Path(path).read_text())
    assert rec["streams"]["rows"] and rec["config"]["lowerings"], "the legs live in the full record"
    assert line["full_record"] == rec["full_record"]


def test_a_line_without_optional_legs_is_still_valid():
    full = full_size_stub()
    for k in ("config4_resnet20_nt65536_N131072", "cpu_baseline", "ks_traffic", "streams", "real_bootstrap"):
        full[k] = None
    line = bench.compact_line(full)
    assert "cpu_baseline" not in line and "config4" not in line and line["roofline"]["frac"]
    json.loads(json.dumps(line))
