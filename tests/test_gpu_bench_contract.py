"""bench.py as the driver runs it: one JSON line on stdout with the fields the contract names.  Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv):
    env = dict(os.environ, SCCBF_NO_REBUILD="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv),
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # exactly one line, and it is JSON
    return json.loads(lines[0])


def _step_line(d, lo=0.02):
    assert d["us_per_step"] > 0 and d["solves_per_s"] > 0 and lo < d["frac"] < 1.0, d


def test_bench_line_has_the_contracts_fields():
    """The command the driver runs: twenty steps = one launch over twenty batch records."""
    d = _bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "1")
    assert d["metric"].startswith("CBF-QP solves/sec") and d["unit"] == "solves/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 65536 * 20 / (d["ms_per_step"] * 20 * 1e-3)) <= 1e-6 * d["value"]
    cfg = d["config"]
    assert cfg["batches_per_launch"] == 20 and cfg["launches_in_timed_region"] == 1
    assert "sccbf_solve_batches_f32" in cfg["launch"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == 20 * 65536 * 548 and r["algorithmic_bytes_per_step"] == 65536 * 548
    assert abs(r["kernel_ms"] - 20 * r["kernel_ms_per_step"]) < 1e-9
    # the host's clock around the region and the device's time of its one kernel tell the same story
    assert 0.8 < r["frac_by_wall_clock"] / r["frac"] <= 1.0
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    assert r["fused_answers_equal_separate_launches"] is True
    assert 0.2 < r["one_at_a_time"]["frac"] < r["frac"] and 0.3 < r["branched_graph"]["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "solves/s" and c["value"] > 0 and c["value_1thread"] > 0
    assert c["cores"] == c["threads"] >= 1 and c["host_cores"] >= 1 and "sample" in c
    assert 0.0 <= d["infeasible_fraction"] <= 1.0 and d["steady_state"]["roofline_frac"] > r["frac"] * 0.8
    # every configuration of the baseline, the lane variants, slots as written, one GPU's share of C4 as stated
    cf = d["configs"]
    _step_line(cf["c2"], lo=0.0)
    _step_line(cf["c3"])
    assert cf["c3"]["with_nominal_controller"]["us_per_step"] > cf["c3"]["us_per_step"]
    for prec in ("f32", "f64"):
        _step_line(cf["c4"][prec])
        _step_line(cf["c5"][prec])
        _step_line(cf["c5"][prec]["one_launch_per_step"])
    _step_line(cf["c4"]["f64"]["one_launch_per_step"])
    for alpha in ("alpha_0.5", "alpha_1", "alpha_5"):
        t = cf["c5"]["fp32_vs_fp64"][alpha]
        assert t["status_agreement"] >= 0.999 and t["max_rel_du"] <= 1e-3 and t["well_conditioned"] >= 0.9 * t["optimal_in_both"]
    assert d["fp64"]["c4"] == cf["c4"]["f64"] and d["fp64"]["c5"] == cf["c5"]["f64"]
    for k in ("auto", "lanes_1", "lanes_2", "lanes_4", "lanes_8", "lanes_16", "lanes_64"):
        _step_line(d["variants"][k]["twenty_batches_per_launch"])
        _step_line(d["variants"][k]["one_launch_per_step"])
    w = d["slots_as_written"]
    _step_line(w["twenty_batches_per_launch"])
    assert d["value_serial_written"] == w["one_launch_per_step"]["solves_per_s"]
    for n in (65536, 32768, 16384, 8192):
        sh = d["c4_shard"]["%d_agents" % n]
        assert sh["gpus_sharing_65536"] == 65536 // n
        _step_line(sh["twenty_steps_per_launch"])
        _step_line(sh["one_launch_per_step"])
    assert 0.5 < d["roofline_large_batch"]["frac"] < 1.0
    # long obstacle lists, from per-agent lists and from tiles
    for M in (128, 257, 1024):
        for layout in ("agents", "tiles"):
            row = d["long_lists"]["rows_%d" % M][layout]
            assert 0.35 < row["frac"] < 1.0 and abs(row["solves_per_s"] * row["us_per_launch"] * 1e-6 - 65536) < 1.0
        assert d["long_lists"]["rows_%d" % M]["agents"]["frac"] > 0.5


def test_a_long_region_and_one_launch_per_step():
    d = _bench("--steps", "400", "--warmup", "40", "--quick", "--no-cpu")
    assert d["config"]["launches_in_timed_region"] == 20
    assert 0.5 < d["roofline"]["frac"] < 1.0
    d1 = _bench("--steps", "40", "--warmup", "5", "--quick", "--no-cpu", "--fuse", "1")
    assert d1["config"]["launches_in_timed_region"] == 40 and d1["config"]["batches_per_launch"] == 1
    assert "sccbf_solve_f32 launch per step" in d1["config"]["launch"]
