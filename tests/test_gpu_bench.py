"""The bench line's contract (task statement): one JSON line with the metric, the roofline and the CPU baseline of the same run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_on_a_small_run():
    d = _run("--config", "B", "--steps", "6", "--warmup", "2", "--no-extra", "--cpu-sample", "65536")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "candidates/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 2048 * 2048 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "device_ms_per_step", "set_phase_ms", "set_phase_hbm_frac", "store_roofline_frac"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["kernel_ms"] < r["device_ms_per_step"] <= d["ms_per_step"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == "candidates/s"
    assert d["config"]["guard_audit_violations"] == 0
