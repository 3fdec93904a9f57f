"""bench.py --gpus N starts N ranks itself: rehearsed here on the CPU (gloo, the solve is a copy)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_and_the_collective_sees_both():
    r = _run(["--gpus", "2", "--cpu-stub", "--steps", "20", "--warmup", "5", "--agents", "256"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gloo_ranks"] == 2 and out["stub"] is True
    assert len(out["ms_per_step_by_rank"]) == 2 and out["gather_ms_per_step"] >= 0.0
    assert out["steps"] == 20 and out["scaling"] == "weak"
    assert out["config"]["agents_per_gpu"] == 256 and out["config"]["agents_per_step_whole_job"] == 512
    # the other partition of the work beside it: one 256-agent batch per step, 128 agents per rank
    o = out["other_scaling"]
    assert o["scaling"] == "strong" and o["agents_per_rank_and_step"] == 128 and o["agents_per_step_whole_job"] == 256
    assert len(o["ms_per_step_by_rank"]) == 2 and o["gather_ms_per_step"] >= 0.0


def test_strong_scaling_line_shards_one_batch_over_the_ranks():
    r = _run(["--gpus", "2", "--cpu-stub", "--steps", "20", "--warmup", "5", "--agents", "320", "--scaling", "strong"])
    assert r.returncode == 0, r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == 2 and out["gloo_ranks"] == 2 and out["scaling"] == "strong"
    # 320 agents = 5 tiles of 64: blocks of whole tiles, rank 0 holds three of them
    assert out["config"]["agents_per_gpu"] == 192 and out["config"]["agents_per_step_whole_job"] == 320
    assert "contiguous blocks" in out["config"]["workload"]
    assert abs(out["value"] - 320 * 20 / (out["ms_per_step"] * 20 * 1e-3)) <= 1e-6 * out["value"]
    assert out["other_scaling"]["scaling"] == "weak" and out["other_scaling"]["agents_per_step_whole_job"] == 640
    r = _run(["--gpus", "2", "--cpu-stub", "--steps", "4", "--warmup", "1", "--agents", "256", "--scaling", "strong",
              "--no-other-scaling"])
    assert r.returncode == 0 and "other_scaling" not in json.loads(r.stdout.strip().splitlines()[-1])


def test_gpus_beyond_the_visible_devices_is_an_error_not_a_smaller_job():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(max(n, 2)), "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "visible" in r.stderr


def test_gpus_must_match_the_launcher_world_size():
    r = _run(["--gpus", "2", "--cpu-stub", "--steps", "2", "--warmup", "1"],
             env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and r.stdout.strip() == ""
