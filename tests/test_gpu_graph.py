"""The entry points inside a captured hipGraph (how bench.py times them and how a caller would run a
tick loop): every replay must give the eager results, with eager work and synchronisations between
the replays."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled  # noqa: E402
from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _batch(seed, A=4096):
    w = scenarios.config4(A=A, seed=seed, nominal="uniform")
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in w.items()}
    t = {k: (v.float().contiguous() if v.dtype.is_floating_point else v) for k, v in t.items()}
    t["obs"] = planes_to_tiled(t["obs"])
    t["u_out"] = torch.empty_like(t["u_ref"])
    t["status"] = torch.empty((A,), dtype=torch.int32, device=DEV)
    return t


def test_filter_replays_equal_eager():
    flt = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]))
    ring = [_batch(s) for s in (11, 12, 13)]
    eager = []
    for b in ring:
        u, st = flt.solve(b["state"], b["u_ref"], b["obs"], b["obs_type"])
        eager.append((u.clone(), st.clone()))
    other = _batch(14, A=1000)
    stream = torch.cuda.Stream(DEV)
    with torch.cuda.stream(stream):
        for b in ring:   # warm-up on the capture stream
            flt.solve(b["state"], b["u_ref"], b["obs"], b["obs_type"], u_out=b["u_out"], status=b["status"])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for b in ring:
                flt.solve(b["state"], b["u_ref"], b["obs"], b["obs_type"], u_out=b["u_out"], status=b["status"])
        for replay in range(4):
            for b in ring:   # eager work between the replays: the outputs are wiped, another batch is solved
                b["u_out"].fill_(float("nan"))
                b["status"].fill_(-1)
            flt.solve(other["state"], other["u_ref"], other["obs"], other["obs_type"])
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for b, (u, st) in zip(ring, eager):
                assert torch.equal(b["u_out"], u) and torch.equal(b["status"], st), replay


def test_closed_loop_ticks_replay_equal_eager():
    g_ = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sce_course.npz"))
    traj = torch.from_numpy(np.stack([g_["cx"], g_["cy"], g_["cyaw"], np.full_like(g_["cx"], 8.0)], axis=1)).float().to(DEV)
    A = 2048
    w = scenarios.config4(A=A, seed=3, nominal="uniform")
    rng = np.random.default_rng(4)
    at = rng.integers(0, traj.shape[0] - 300, A)
    xy = np.stack([g_["cx"][at] + rng.normal(0, 2, A), g_["cy"][at] + rng.normal(0, 2, A)], axis=1)
    w["obs"][:, 0, :, 0:2] += (xy - w["state"][:, 0:2])[None, :, :]
    w["state"][:, 0:2] = xy
    w["state"][:, 2] = g_["cyaw"][at]
    state0 = torch.from_numpy(w["state"]).float().to(DEV)
    obs = planes_to_tiled(torch.from_numpy(w["obs"]).float().to(DEV))
    ty = torch.from_numpy(w["obs_type"]).to(DEV)

    def loop_on(state):
        return ClosedLoop(BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5])), BatchedNominal(axle=2.9, k=0.5, ks=0.01, kp=1.0, ki=0.1),
                          traj, state, obs, ty, use_pid_state=True)
    T = 6
    ref = loop_on(state0.clone())
    for _ in range(T):
        ref.tick()
    torch.cuda.synchronize()
    loop = loop_on(state0.clone())
    stream = torch.cuda.Stream(DEV)
    with torch.cuda.stream(stream):
        loop.tick()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            loop.state.copy_(state0)          # every replay starts from the same states
            loop.target_idx.fill_(0)
            loop.pid_state.fill_(0)
            for _ in range(T):
                loop.tick()
        for replay in range(3):
            loop.state.fill_(float("nan"))   # eager work between the replays
            loop.u.fill_(0)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for name in ("state", "u", "u_ref", "status", "beta", "target_idx", "pid_state"):
                assert torch.equal(getattr(loop, name), getattr(ref, name)), (replay, name)


def test_rollout_replays_equal_eager():
    g_ = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sce_course.npz"))
    traj = torch.from_numpy(np.stack([g_["cx"], g_["cy"], g_["cyaw"], np.full_like(g_["cx"], 8.0)], axis=1)).float().to(DEV)
    A, T = 1000, 12
    w = scenarios.config4(A=A, seed=8, nominal="uniform")
    rng = np.random.default_rng(9)
    at = rng.integers(0, traj.shape[0] - 300, A)
    xy = np.stack([g_["cx"][at] + rng.normal(0, 2, A), g_["cy"][at] + rng.normal(0, 2, A)], axis=1)
    w["obs"][:, 0, :, 0:2] += (xy - w["state"][:, 0:2])[None, :, :]
    w["state"][:, 0:2] = xy
    w["state"][:, 2] = g_["cyaw"][at]
    state0 = torch.from_numpy(w["state"]).float().to(DEV)
    loop = ClosedLoop(BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5])), BatchedNominal(axle=2.9, k=0.5, ks=0.01), traj,
                      state0.clone(), torch.from_numpy(w["obs"]).float().to(DEV), torch.from_numpy(w["obs_type"]).to(DEV))
    ref = loop.rollout(T, motion="constant_velocity", trace=True)
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in ref.items()}
    ref_state = loop.state.clone()
    out = {k: torch.empty_like(v) for k, v in ref.items()}
    stream = torch.cuda.Stream(DEV)
    with torch.cuda.stream(stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            loop.state.copy_(state0)
            loop.target_idx.fill_(0)
            loop.rollout(T, motion="constant_velocity", out=out)
        for replay in range(3):
            for v in out.values():
                v.fill_(0)
            loop.state.fill_(0)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(loop.state, ref_state), replay
            for k in ref:
                assert torch.equal(out[k], ref[k]), (replay, k)


def test_prewarm_lets_a_first_long_list_launch_be_captured():
    """A fresh process whose FIRST launch with rows in LDS only (300 rows: tables beyond 64 KB per workgroup) and
    whose first sccbf_solve_batches_static happen inside stream capture: refused without sccbf_prewarm, fine after it,
    and the replays give the eager answers."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
from sccav_cbf_amd import _lib, scenarios
from sccav_cbf_amd.batched import BatchedCBF, BatchTable, prewarm
dev = torch.device("cuda:0")
w = scenarios.monte_carlo(200, 150, 150, 3, nominal="uniform")
t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in w.items()}
t = {k: (v.float().contiguous() if v.dtype.is_floating_point else v) for k, v in t.items()}
t["u_out"] = torch.empty_like(t["u_ref"]); t["status"] = torch.empty((200,), dtype=torch.int32, device=dev)
flt = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]))
table = BatchTable([t])
stream = torch.cuda.Stream(dev)
def capture():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        flt.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], u_out=t["u_out"], status=t["status"])
        flt.solve_batches_static(table)
    return g
with torch.cuda.stream(stream):
    if "--no-prewarm" in sys.argv:
        try:
            capture()
        except Exception as e:
            print("REFUSED", type(e).__name__)
            sys.exit(0)
        print("CAPTURED WITHOUT PREWARM")
        sys.exit(0)
    prewarm()
    g = capture()
    for _ in range(3):
        t["u_out"].fill_(float("nan")); t["status"].fill_(-1)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        got = (t["u_out"].clone(), t["status"].clone())
        u, st = flt.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"])
        torch.cuda.synchronize()
        assert torch.equal(got[0], u) and torch.equal(got[1], st)
print("OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCCBF_NO_REBUILD="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout[-500:] + res.stderr[-1500:]
