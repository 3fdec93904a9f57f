"""bench.py as the driver runs it: one JSON line on stdout with the fields the contract names.  Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contracts_fields():
    env = dict(os.environ, SCCBF_NO_REBUILD="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--no-large", "--no-fp64", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly one line, and it is JSON
    d = json.loads(lines[0])
    assert d["metric"].startswith("CBF-QP solves/sec") and d["unit"] == "solves/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 65536 * 20 / (d["ms_per_step"] * 20 * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.2 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == 65536 * 548
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    assert r["overlapped_answers_equal_serial"] is True and 0.2 < r["one_at_a_time"]["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "solves/s" and c["value"] > 0 and c["value_1thread"] > 0
    assert c["cores"] == c["threads"] >= 1 and c["host_cores"] >= 1 and "sample" in c
    assert 0.0 <= d["infeasible_fraction"] <= 1.0 and d["steady_state"]["roofline_frac"] > r["frac"] * 0.9
