"""bench.py's output contract: ONE JSON line on stdout with the metric of BASELINE.json, the roofline object of the
dominant kernel and the CPU baseline (small workload here; the default run is the 200^3 configuration)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cells", "40"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(" ")[0] == base["metric"].split(" ")[0] == "DOF/s" and d["unit"] == "DOF/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 2, 1)
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["free_dofs"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["converged_reason"] == 2 and d["max_nodal_error"] < 1e-3
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    assert rf["algorithmic_bytes_per_launch"] == 12 * rf["nnz"] + 20 * rf["rows"] and rf["launches_timed"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "DOF/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
