"""The reference-compatible object API driving the HIP path: closed loops C1 and C2."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import replay  # noqa: E402
from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.cbf._vec import Point2, Vector2  # noqa: E402
from sccav_cbf_amd.cbf.cbf import DBM_CBF_2DS, KBM_VC_CBF2D  # noqa: E402
from sccav_cbf_amd.cbf.obstacles import (CollisionCone2D, Ellipse2D, ObstacleList2D, PolyLane,  # noqa: E402
                                         RadialEllipse2D)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_c1_closed_loop_through_object_api_matches_mat():
    """The driver's call pattern (stanley_controller_ellipse.py:733-747) on the GPU path must
    reproduce the reference's saved slip-angle history to 1e-3 degree."""
    k = {}

    def gpu_tick(s, u_ref, slot, kind):
        cbf = DBM_CBF_2DS(alpha=1)
        cbf.set_model_params(lr=replay.LR, lf=replay.LF)
        s_obs = np.array([slot[0], slot[1], 0.0, 0.0])
        cbf.obstacle_list2d.update({0: CollisionCone2D(k["a_cone"], s, s_obs)})
        cbf.update_state(s=s)
        cbf.set_qp_cost_weight(np.diag([0.5, 0.5]))
        u = cbf.solve_cbf(np.array(u_ref))
        return u[0], u[1]

    cx, cy, _ = replay.load_course()
    k.update(replay.scenario_constants(cx, cy))
    t, beta_deg, log = replay.replay_stanley_cone(solve_tick=gpu_tick)
    g = np.load(os.path.join(GOLDEN, "beta_vs_time.npz"))
    assert len(t) == 277
    assert np.abs(beta_deg - g["beta_deg"]).max() <= 1e-3
    # and tick by tick against the oracle run
    _, beta_orc, _ = replay.replay_stanley_cone()
    assert np.abs(beta_deg - beta_orc).max() <= 1e-6


def test_object_api_partials_and_rows(oracle):
    s = np.array([1.0, 2.0, 0.4, 6.0])
    ol = ObstacleList2D()
    ol[0] = Ellipse2D(6, 3, Point2(20, 4), theta=0.3, buffer=0.5)
    ol[0].update_velocity_by_magnitude(1.5)
    ol[1] = CollisionCone2D(3.0, s, [25, -3, 2.0, 1.0])
    ol[2] = PolyLane(np.array([9.0, 0.05, 0.002, -1e-4]), s=s)
    ol[3] = RadialEllipse2D(1.7, 1.7, center=Vector2(12, 9), kv=1.0)
    ol[3].update_velocity(Vector2(-2.0, -1.0))
    ol[4] = PolyLane(np.array([-6.0, 0.1, -0.004, 2e-4, 3e-6, -2e-7, 1e-9]), s=s, buffer=1.0)   # degree 6
    ol.update_state(s=s)
    kinds, slots = ol.slots()
    assert list(kinds) == [0, 1, 2, 3, 8]
    exp = [oracle.barrier(int(kinds[i]), s, slots[i]) for i in range(5)]
    for name, key in (("f", "h"), ("dx", "hx"), ("dy", "hy"), ("dtheta", "hth"), ("dv", "hv"), ("dt", "ht")):
        got = getattr(ol, name)()
        assert got.shape == (5, 1)
        assert np.allclose(got[:, 0], [e[key] for e in exp], rtol=1e-10, atol=1e-10), name
        for i, o in enumerate(ol.values()):
            assert abs(getattr(o, name)() - exp[i][key]) <= 1e-10 * (1 + abs(exp[i][key]))
    assert np.allclose(ol.gradient(), [[e["hx"], e["hy"], e["hth"], e["hv"]] for e in exp], atol=1e-10)
    assert ol[1].dbeta() == ol[1].dtheta()
    # stacked rows and the QP through the filter object
    cbf = DBM_CBF_2DS(alpha=0.8)
    cbf.set_model_params(lr=1.45, lf=1.45)
    cbf.obstacle_list2d = ol
    cbf.update_state(s=s)
    cbf.set_qp_cost_weight(np.diag([0.5, 0.5]))
    info, u = cbf.solve_cbf(np.array([0.5, 0.05]), return_solver=True)
    p = oracle.make_params(alpha=0.8, R=(0.5, 0, 0, 0.5))
    obs = np.ascontiguousarray(slots.reshape(5, 2, 1, 4))
    uo, so, ro = oracle.solve_batch(p, s[None], np.array([[0.5, 0.05]]), obs, kinds, want_rows=True)
    assert info["status_code"] == so[0]
    assert np.allclose(info["rows"], ro[:, 0, :], rtol=1e-9, atol=1e-9)
    assert np.allclose(u, uo[0], atol=1e-9) and np.allclose(info["x"], u)


def test_kbm_object_api(oracle):
    cbf = KBM_VC_CBF2D(alpha=1.0)
    cbf.set_model_params(L=2.9)
    cbf.obstacle_list2d.update({0: Ellipse2D(20, 10, Point2(30.0, 2.0), buffer=2.0)})
    cbf.update_state(Vector2(0.0, 0.0), theta=0.05)
    info, u = cbf.solve_cbf(np.array([8.0, 0.02]))
    p = oracle.make_params(alpha=1.0, dynamics=oracle.DYN_KBM)
    kinds, slots = cbf.obstacle_list2d.slots()
    uo, so, _ = oracle.solve_batch(p, np.array([[0, 0, 0.05, 0.0]]), np.array([[8.0, 0.02]]),
                                   np.ascontiguousarray(slots.reshape(1, 2, 1, 4)), kinds)
    assert info["status_code"] == so[0] == 0
    assert np.allclose(u, uo[0], atol=1e-9)
    assert u[0] < 8.0        # the filter brakes in front of the ellipse


def test_c2_radial_closed_loop(oracle):
    """1 ego x 32 radial seekers, fp64, 600 ticks at 30 Hz (radial_dynamic_obstacles.py
    427-507 extended to 32 obstacles): every tick on the GPU path (one wavefront, rows in
    LDS) against the oracle, with the seekers and the ego stepped as the script does."""
    from sccav_cbf_amd.batched import BatchedCBF
    dev = torch.device("cuda:0")
    w = scenarios.radial_seekers(M=32)
    M = 32
    dt = 1.0 / 30.0
    lr = lf = 1.45
    p = oracle.make_params(alpha=1.0, lr=lr, lf=lf, R=(1, 0, 0, 1))
    flt = BatchedCBF(alpha=1.0, lr=lr, lf=lf, R=np.eye(2), lanes_per_qp=64)
    s = np.zeros(4)
    slots = np.ascontiguousarray(w["obs"].transpose(0, 2, 1, 3).reshape(M, 8))
    kinds = torch.from_numpy(w["obs_type"]).to(dev)
    worst, modified, infeasible = 0.0, 0, 0
    for tick in range(600):
        obs = np.ascontiguousarray(slots.reshape(M, 2, 1, 4))
        uo, so, _ = oracle.solve_batch(p, s[None], np.zeros((1, 2)), obs, w["obs_type"])
        ug, sg = flt.solve(torch.from_numpy(s[None].copy()).to(dev), torch.zeros((1, 2), dtype=torch.float64, device=dev),
                           torch.from_numpy(obs).to(dev), kinds)
        ug, sg = ug.cpu().numpy(), sg.cpu().numpy()
        assert sg[0] == so[0], tick
        worst = max(worst, np.abs(ug - uo).max())
        modified += int(np.abs(ug[0]).max() > 1e-9)
        infeasible += int(sg[0] != 0)
        oracle.vehicle_step(s, ug[0, 0], ug[0, 1], lr, lf, np.radians(30.0), dt)
        # seekers: head for the ego at max(0.2 dist, 3) m/s (radial_dynamic_obstacles.py:193-239)
        dx, dy = s[0] - slots[:, 0], s[1] - slots[:, 1]
        dist = np.hypot(dx, dy)
        yaw = np.arctan2(dy, dx)
        sp = np.maximum(0.2 * dist, 3.0)
        slots[:, 4], slots[:, 5] = sp * np.cos(yaw), sp * np.sin(yaw)
        slots[:, 0] += slots[:, 4] * dt
        slots[:, 1] += slots[:, 5] * dt
    assert worst <= 1e-8
    assert modified > 10
    print("C2: ticks with intervention %d, infeasible %d, worst |du| %.2e" % (modified, infeasible, worst))


def test_independent_launches_in_parallel_graph_branches(oracle):
    """batched.capture_independent: the batches of a sweep captured on three parallel branches of one hipGraph give
    the answers of the same launches one at a time, on every replay, with eager work and synchronisations between
    the replays."""
    from sccav_cbf_amd import scenarios
    from sccav_cbf_amd.batched import BatchedCBF, capture_independent, planes_to_tiled
    dev = torch.device("cuda:0")
    flt = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), streamed_obs=True)
    batches = []
    for b in range(7):
        w = scenarios.prepared(scenarios.config4(A=4096 + 64 * b, nominal="uniform", seed=40 + b))
        t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in w.items()}
        t = {k: (v.float() if v.dtype.is_floating_point else v) for k, v in t.items()}
        t["obs"] = planes_to_tiled(t["obs"])
        A = t["state"].shape[0]
        t["u"] = torch.zeros((A, 2), device=dev)
        t["st"] = torch.full((A,), -1, dtype=torch.int32, device=dev)
        batches.append(t)

    def launch(t):
        return lambda: flt.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], u_out=t["u"], status=t["st"])
    steps = [launch(t) for t in batches]
    for f in steps:
        f()
    torch.cuda.synchronize()
    want = [(t["u"].clone(), t["st"].clone()) for t in batches]
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        g = capture_independent(steps, branches=3)
        for rep in range(4):
            for t in batches:                      # eager work between the replays: the outputs are wiped
                t["u"].fill_(float(rep))
                t["st"].fill_(-7)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for (u, st), t in zip(want, batches):
                assert torch.equal(u, t["u"]) and torch.equal(st, t["st"]), rep


def test_object_api_in_a_process_that_never_imported_torch_itself():
    """A driver script in the reference's style imports the cbf classes and numpy, not torch.  The library must
    then still end up on the HIP runtime of the tensors it is handed (PyTorch's own copy): loaded ahead of torch it
    would map the system's runtime and no kernel would launch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from sccav_cbf_amd import _lib\n"
        "assert 'torch' not in sys.modules\n"
        "_lib.load()\n"
        "from sccav_cbf_amd.cbf.cbf import DBM_CBF_2DS\n"
        "from sccav_cbf_amd.cbf.obstacles import CollisionCone2D\n"
        "s = np.array([0.0, 5.0, 0.35, 10.0])\n"
        "cbf = DBM_CBF_2DS(alpha=1)\n"
        "cbf.set_model_params(lr=1.45, lf=1.45)\n"
        "cbf.obstacle_list2d.update({0: CollisionCone2D(11.18, s, np.array([30.0, 12.0, 0.0, 0.0]))})\n"
        "cbf.update_state(s=s)\n"
        "cbf.set_qp_cost_weight(np.diag([0.5, 0.5]))\n"
        "u = cbf.solve_cbf(np.array([0.3, 0.1]))\n"
        "print('U', repr(float(u[0])), repr(float(u[1])))\n" % root)
    env = dict(os.environ, SCCBF_NO_REBUILD="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    got = [float(x) for x in [l for l in res.stdout.splitlines() if l.startswith("U ")][0].split()[1:]]
    s = np.array([0.0, 5.0, 0.35, 10.0])
    cbf = DBM_CBF_2DS(alpha=1)
    cbf.set_model_params(lr=1.45, lf=1.45)
    cbf.obstacle_list2d.update({0: CollisionCone2D(11.18, s, np.array([30.0, 12.0, 0.0, 0.0]))})
    cbf.update_state(s=s)
    cbf.set_qp_cost_weight(np.diag([0.5, 0.5]))
    u = cbf.solve_cbf(np.array([0.3, 0.1]))
    assert abs(got[0] - float(u[0])) <= 1e-12 and abs(got[1] - float(u[1])) <= 1e-12
    assert abs(got[0] - 0.3) > 1e-6 or abs(got[1] - 0.1) > 1e-6      # the cone is in the way: the filter acts


def test_object_api_least_violation_policy(oracle):
    """Two circles that leave no admissible input between them: with infeasible_policy = "min_violation" solve_cbf
    returns the oracle's least-violation input instead of the nominal one; last_status stays 1."""
    import warnings
    from sccav_cbf_amd.cbf.cbf import CBFStatusWarning, SADBM_CBF_2DS
    s = np.array([0.0, 0.0, 0.0, 8.0])
    cbf = DBM_CBF_2DS(alpha=1.0)
    cbf.set_model_params(lr=1.45, lf=1.45)
    cbf.set_qp_cost_weight(np.diag([0.5, 0.5]))
    cbf.obstacle_list2d.update({0: Ellipse2D(4, 4, Point2(6.0, 2.5)), 1: Ellipse2D(4, 4, Point2(6.0, -2.5)),
                                2: CollisionCone2D(3.0, s, [9.0, 0.0, np.pi, 4.0])})
    cbf.update_state(s=s)
    u_ref = np.array([0.5, 0.02])
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        info, u_nom = cbf.solve_cbf(u_ref, return_solver=True)
    assert cbf.last_status == 1 and np.array_equal(u_nom, u_ref) and "nominal input unchanged" in str(rec[-1].message)
    cbf.infeasible_policy = "min_violation"
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        u = cbf.solve_cbf(u_ref)
    assert cbf.last_status == 1 and issubclass(rec[-1].category, CBFStatusWarning) and "least" in str(rec[-1].message)
    rows = info["rows"]
    beta_ref = np.arctan2(1.45 * np.tan(u_ref[1]), 2.9)
    uo, so, t = oracle.qp_min_violation_batch(oracle.make_params(R=(0.5, 0, 0, 0.5)), rows[:, None, :],
                                              np.array([[u_ref[0], beta_ref]]))
    assert so[0] == 1 and t[0] > 0
    got = np.array([u[0], np.arctan2(1.45 * np.tan(u[1]), 2.9)])
    assert np.abs(got - uo[0]).max() <= 1e-7 * (1 + np.abs(uo[0]).max())
    sa = SADBM_CBF_2DS(alpha=1.0, dt=0.1)
    sa.infeasible_policy = "min_violation"
    sa.obstacle_list2d.update({0: Ellipse2D(4, 4, Point2(6.0, 2.5))})
    sa.update_state(s=s)
    with pytest.raises(ValueError, match="nominal"):
        sa.solve_cbf(u_ref)
