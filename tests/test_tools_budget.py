"""CPU: the budget summarisers (tools/summarize/per_op_budget.py, kernel_bytes.py) on a committed sample of rocprofv3 output -- the four passes of
`tools/legs/per_op_only.py 20 --only rescale` of round 5 (tests/golden/rocprof_sample/: kernel trace, FETCH_SIZE, WRITE_SIZE, VALU counters;
gzip of the profiler's own CSVs).  The tables under profiles/ come out of these scripts; this keeps them runnable and their arithmetic fixed:
bytes = FETCH_SIZE x 2 (the gfx950 correction) + WRITE_SIZE in KB, VALU floor = instructions x 4 cycles / 1024 SIMDs / 2.05 GHz."""
import gzip
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SAMPLE = ROOT / "tests" / "golden" / "rocprof_sample"


def _unpack(tmp_path):
    out = {}
    for n in ("kernel_trace", "fetch", "write", "valu"):
        p = tmp_path / f"{n}.csv"
        p.write_bytes(gzip.open(SAMPLE / f"{n}.csv.gz").read())
        out[n] = str(p)
    return out


def test_per_op_budget_reproduces_the_rescale_table(tmp_path):
    f = _unpack(tmp_path)
    js = tmp_path / "b.json"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "summarize" / "per_op_budget.py"), "rescale", "20", f["kernel_trace"], f["fetch"], f["write"], f["valu"],
                        "event_us=23.0", f"json={js}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    d = json.loads(js.read_text())
    ks = d["kernels"]
    assert [k["kernel"].split("<")[0] for k in ks] == ["f_irows_rs_kernel", "f_dr_icols_lift_fcols_kernel", "f_frows_final_kernel"]   # launch order
    assert all(k["launches_per_op"] == 1 for k in ks) and d["iterations"] == 20
    mid = ks[1]
    # 2 polynomials x 12 targets x 64 tiles of 512 coefficients; bytes: it writes its 24 limbs of 256 KiB and reads the 2 dropped limbs
    assert mid["workgroups"] == 1536 and 6.2 < mid["write_MB"] < 6.4 and mid["read_MB"] < 1.0
    assert abs(mid["floor_valu_us"] - mid["valu_wave_instructions"] * 4 / 1024 / 2.05e9 * 1e6) < 0.02
    assert abs(mid["floor_bytes_us"] - (mid["read_MB"] + mid["write_MB"]) * 1e6 / 5.5e12 * 1e6) < 0.02
    assert mid["floor_us"] == max(mid["floor_bytes_us"], mid["floor_valu_quantised_us"], 3.7)
    assert 15.0 < d["kernel_time_per_op_us"] < 30.0 and d["sum_of_floors_us"] < d["kernel_time_per_op_us"]
    assert "== rescale: 3 kernels per op" in r.stdout


def test_kernel_bytes_sums_the_same_counters(tmp_path):
    f = _unpack(tmp_path)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "summarize" / "kernel_bytes.py"), f["kernel_trace"], f["fetch"], f["write"], "top=5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("f_dr_icols_lift_fcols_kernel")]
    assert len(rows) == 1 and " 21 " in rows[0]   # the warm-up call + 20 timed ones
    assert "sum of the kernels listed" in r.stdout
