"""The unicycle-input (DUM_CBF_2DS) and steer-rate (SADBM_CBF_2DS) models on the GPU against the
oracle, through the C ABI and through the object API.  Same bars as tests/test_gpu_parity.py."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.cbf._vec import Point2  # noqa: E402
from sccav_cbf_amd.cbf.cbf import DUM_CBF_2DS, SADBM_CBF_2DS  # noqa: E402
from sccav_cbf_amd.cbf.obstacles import CollisionCone2D, Ellipse2D  # noqa: E402
from tests.test_gpu_parity import R_HALF, _check, _dev, _filter, _to_dev  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("lanes", [0, 4, 8, 16])
@pytest.mark.parametrize("name", ["mixed16", "lanes2_cones6", "lanes6_cones6"])
def test_dum_matches_oracle(oracle, name, dtype, lanes):
    w = (scenarios.config4(A=4096, nominal="uniform") if name == "mixed16"
         else scenarios.lanes_and_cones(A=4096, curved=(name == "lanes6_cones6")))
    w["u_ref"][:, 1] *= 3.0     # yaw rates, not steering angles
    p = oracle.make_params(alpha=1.0, R=R_HALF.reshape(-1), dynamics=oracle.DYN_DUM)
    uo, so, ro = oracle.solve_batch(p, w["state"], w["u_ref"], w["obs"], w["obs_type"], want_rows=True)
    f = _filter(lanes=lanes, dynamics="dum")
    t = _to_dev(w, dtype)
    ug, sg, rg = f.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], want_rows=True)
    torch.cuda.synchronize()
    ug, sg, rg = ug.double().cpu().numpy(), sg.cpu().numpy(), rg.double().cpu().numpy()
    _check(ug, sg, uo, so, ro, w["u_ref"], dtype)
    tol = 1e-9 if dtype == torch.float64 else 1e-3
    assert (np.abs(rg - ro) / (np.abs(ro).max(axis=2, keepdims=True) + 1.0)).max() <= tol
    assert (so == 0).sum() > 1000 and (np.abs(uo - w["u_ref"]).max(axis=1) > 1e-3).sum() > 300


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("lanes,cone_beta_state", [(0, True), (4, False), (8, False), (16, True)])
def test_sadbm_matches_oracle_over_ticks(oracle, dtype, lanes, cone_beta_state):
    """Four ticks of SADBM_CBF_2DS.solve_cbf (cbf/cbf.py:349-437) for 4096 agents: each tick starts
    from the oracle's slip state, the agents move by the oracle's vehicle step in between."""
    A, dt = 4096, 0.1
    w = scenarios.config4(A=A, nominal="uniform")
    flags = oracle.FLAG_CONE_BETA_STATE if cone_beta_state else 0
    p = oracle.make_params(alpha=1.0, R=R_HALF.reshape(-1), dynamics=oracle.DYN_SADBM, flags=flags)
    f = _filter(lanes=lanes, dynamics="sadbm")
    t = _to_dev(w, dtype)
    state = w["state"].copy()
    slip = np.zeros((A, 2))
    rng = np.random.default_rng(3)
    acted = 0
    for tick in range(4):
        u_ref = w["u_ref"] + rng.normal(0, 0.02, (A, 2))
        before = slip.copy()
        uo, so, ro = oracle.solve_sa_batch(p, dt, state, slip, u_ref, w["obs"], w["obs_type"], want_rows=True)
        sl = torch.from_numpy(before).to(_dev()).to(dtype)
        ug, sg, rg = f.solve(torch.from_numpy(state).to(_dev()).to(dtype), torch.from_numpy(u_ref).to(_dev()).to(dtype),
                             t["obs"], t["obs_type"], want_rows=True, slip=sl, dt=dt,
                             cone_beta_state=cone_beta_state)
        torch.cuda.synchronize()
        ug, sg, rg, slg = ug.double().cpu().numpy(), sg.cpu().numpy(), rg.double().cpu().numpy(), sl.double().cpu().numpy()
        # QP coordinates: [a, d(beta)/dt]
        beta_ref = np.arctan2(1.45 * np.tan(u_ref[:, 1]), 2.9)
        un = np.stack([u_ref[:, 0], (beta_ref - before[:, 1]) / dt], axis=1)
        uo_int = np.stack([uo[:, 0], (slip[:, 0] - before[:, 0]) / dt], axis=1)
        ug_int = np.stack([ug[:, 0], (slg[:, 0] - before[:, 0]) / dt], axis=1)
        if dtype == torch.float64:
            assert np.array_equal(sg, so)
            scale = np.maximum(1.0, np.abs(uo_int).max(axis=1))
            assert (np.abs(ug_int - uo_int).max(axis=1) / scale).max() <= 1e-8
            ok = np.abs(slip[:, 0]) < 1.2                       # tan is ill-conditioned near pi/2
            assert np.abs(ug - uo)[ok].max() <= 1e-8 and np.abs(slg - slip).max() <= 1e-9
            assert (np.abs(rg - ro) / (np.abs(ro).max(axis=2, keepdims=True) + 1.0)).max() <= 1e-9
        else:
            _check(ug_int, sg, uo_int, so, ro, un, dtype)
            assert np.abs(slg[:, 1] - slip[:, 1]).max() <= 1e-6
        acted += int(((so == 0) & (np.abs(uo_int - un).max(axis=1) > 1e-3)).sum())
        oracle.vehicle_step_batch(state, uo, 1.45, 1.45, np.radians(30.0), dt)
    assert acted > 1000


def test_sadbm_rejects_what_it_cannot_do():
    f = _filter(dynamics="sadbm")
    w = scenarios.config3(A=64)
    t = _to_dev(w, torch.float64)
    with pytest.raises(ValueError):
        f.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"])                   # no slip state
    g = _filter(dynamics="dbm")
    with pytest.raises(ValueError):
        g.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], slip=torch.zeros((64, 2), device=_dev()), dt=0.1)


def test_object_api_dum_and_sadbm(oracle):
    s = np.array([1.0, 2.0, 0.3, 7.0])
    s_obs = [22.0, 6.0, 2.5, 2.0]

    def obstacles(cbf):
        cbf.obstacle_list2d.update({0: CollisionCone2D(3.0, s, s_obs),
                                    1: Ellipse2D(6, 3, Point2(25.0, 9.0), theta=0.4, buffer=0.5)})
        cbf.update_state(s=s)
        cbf.set_qp_cost_weight(np.diag([0.5, 0.5]))
        return cbf.obstacle_list2d.slots()

    dum = DUM_CBF_2DS(alpha=0.7)
    kinds, slots = obstacles(dum)
    assert dum.gc().shape == (4, 2) and dum.fc().shape == (4, 1)
    info, u = dum.solve_cbf(np.array([0.5, 0.2]), return_solver=True)
    p = oracle.make_params(alpha=0.7, R=(0.5, 0, 0, 0.5), dynamics=oracle.DYN_DUM)
    obs = np.ascontiguousarray(slots.reshape(2, 2, 1, 4))
    uo, so, ro = oracle.solve_batch(p, s[None], np.array([[0.5, 0.2]]), obs, kinds, want_rows=True)
    assert info["status_code"] == so[0] == 0 and np.allclose(u, uo[0], atol=1e-9)
    assert np.allclose(info["rows"], ro[:, 0, :], rtol=1e-9, atol=1e-9)
    assert np.abs(u - [0.5, 0.2]).max() > 1e-3                 # the cone is in the way

    sa = SADBM_CBF_2DS(alpha=0.7, dt=0.05)
    sa.set_model_params(lr=1.4, lf=1.5)
    kinds, slots = obstacles(sa)
    assert sa.gc().shape == (5, 2) and sa.fc().shape == (5, 1)
    p = oracle.make_params(alpha=0.7, lr=1.4, lf=1.5, R=(0.5, 0, 0, 0.5), dynamics=oracle.DYN_SADBM)
    slip = np.zeros((1, 2))
    for tick, u_ref in enumerate(([0.5, 0.05], [0.4, 0.08], [0.3, 0.02])):
        kinds, slots = sa.obstacle_list2d.slots()              # cones carry the slip angle of the last solve
        assert abs(slots[0, 5] - slip[0, 0]) <= 1e-12
        uo, so, _ = oracle.solve_sa_batch(p, 0.05, s[None], slip, np.array([u_ref]),
                                          np.ascontiguousarray(slots.reshape(2, 2, 1, 4)), kinds)
        info, u = sa.solve_cbf(np.array(u_ref), return_solver=True)
        assert info["status_code"] == so[0] == 0
        assert np.allclose(u, uo[0], atol=1e-9)
        assert abs(sa._beta - slip[0, 0]) <= 1e-12 and abs(sa.beta_ref_last - slip[0, 1]) <= 1e-12
        assert sa.obstacle_list2d[0].beta == sa._beta
    empty = SADBM_CBF_2DS(dt=0.1)
    with pytest.raises(ValueError):
        empty.solve_cbf(np.array([0.0, 0.1]))
    assert abs(empty.beta_ref_last - np.arctan2(1.45 * np.tan(0.1), 2.9)) < 1e-15   # cbf.py:369 runs before the raise
    auto = SADBM_CBF_2DS(dt=None)                                # wall-clock mode: dt >= ZERO_TOL
    obstacles(auto)
    auto.solve_cbf(np.array([0.2, 0.01]))
    assert auto._dt >= 1e-3


def test_closed_loop_ticks_with_the_steer_rate_filter(oracle):
    """nominal -> SADBM filter -> vehicle step for 8 ticks on the GPU against the same loop by the oracle."""
    from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
    A, dt = 512, 0.1
    w = scenarios.config4(A=A, nominal="uniform")
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "sce_course.npz"))
    traj = np.ascontiguousarray(np.stack([g["cx"], g["cy"], g["cyaw"], np.full_like(g["cx"], 8.0)], axis=1))
    rng = np.random.default_rng(5)
    at = rng.integers(0, traj.shape[0] - 300, A)
    xy = np.stack([traj[at, 0] + rng.normal(0, 2, A), traj[at, 1] + rng.normal(0, 2, A)], axis=1)
    w["obs"][:, 0, :, 0:2] += (xy - w["state"][:, 0:2])[None, :, :]
    w["state"][:, 0:2] = xy
    w["state"][:, 2] = traj[at, 2] + rng.normal(0, 0.2, A)
    w["state"][:, 3] = rng.uniform(3, 12, A)
    dev = _dev()
    loop = ClosedLoop(_filter(dynamics="sadbm"), BatchedNominal(axle=2.9, k=0.5, ks=0.01, kp=1.0, dt=dt),
                      torch.from_numpy(traj).to(dev), torch.from_numpy(w["state"].copy()).to(dev),
                      torch.from_numpy(w["obs"]).to(dev), torch.from_numpy(w["obs_type"]).to(dev), dt=dt)
    p = oracle.make_params(alpha=1.0, R=R_HALF.reshape(-1), dynamics=oracle.DYN_SADBM, flags=oracle.FLAG_CONE_BETA_STATE)
    s, slip, tgt = w["state"].copy(), np.zeros((A, 2)), np.zeros(A, dtype=np.int32)
    for tick in range(8):
        loop.tick()
        torch.cuda.synchronize()
        ur = oracle.nominal_batch(s, traj, tgt, None, axle=2.9, k=0.5, ks=0.01, kp=1.0, dt=dt)
        uo, so, _ = oracle.solve_sa_batch(p, dt, s, slip, ur, w["obs"], w["obs_type"])
        oracle.vehicle_step_batch(s, uo, 1.45, 1.45, np.radians(30.0), dt)
        sg = loop.status.cpu().numpy()
        same = sg == so
        assert same.mean() >= 0.995
        # agents thrown far off by a vertex on nearly parallel rows amplify the last bits tick by tick
        calm = same & (np.abs(uo).max(axis=1) < 50)
        assert np.abs(loop.u.cpu().numpy() - uo)[calm & (np.abs(slip[:, 0]) < 1.2)].max() <= 1e-6, tick
        assert np.abs(loop.slip.cpu().numpy() - slip)[calm].max() <= 1e-6
        assert np.abs(loop.state.cpu().numpy() - s)[calm].max() <= 1e-6 * (1 + np.abs(s[calm]).max())
        # keep the two loops on the same track where a verdict differed
        loop.state.copy_(torch.from_numpy(s).to(dev))
        loop.slip.copy_(torch.from_numpy(slip).to(dev))
        loop.target_idx.copy_(torch.from_numpy(tgt).to(dev))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", ["dum", "sadbm"])
@pytest.mark.parametrize("M,lanes", [(300, 0), (24, 64), (1024, 64), (24, 32), (300, 16), (700, 0), (1024, 0),
                                     (40, 4), (48, 8), (72, 8), (96, 16), (80, 16), (100, 0)])
def test_one_wavefront_per_qp_for_the_other_models(oracle, dyn, M, lanes, dtype):
    """More rows than sixteen lanes hold (and the same kernel forced on a short list): 16 or 32 lanes or one
    wavefront per QP, rows in LDS only, for the unicycle-input and the steer-rate model; and row counts that
    leave whole register rows of the 4-, 8- and 16-lane kernels unused (their arithmetic is skipped)."""
    A, dt = 96, 0.05
    base = scenarios.config4(A=A, nominal="uniform", seed=5)
    reps = -(-M // 16)
    rng = np.random.default_rng(M)
    obs = np.concatenate([base["obs"]] * reps, axis=0)[:M].copy()
    obs[:, 0, :, 0:2] += rng.normal(0, 6.0, (M, A, 2))            # the copies of an obstacle do not coincide
    ty = np.tile(base["obs_type"], reps)[:M].copy()
    u_ref = base["u_ref"].copy()
    dcode = oracle.DYN_DUM if dyn == "dum" else oracle.DYN_SADBM
    p = oracle.make_params(alpha=1.0, R=R_HALF.reshape(-1), dynamics=dcode)
    f = _filter(lanes=lanes, dynamics=dyn)
    dev = _dev()
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dtype)   # noqa: E731
    if dyn == "dum":
        uo, so, ro = oracle.solve_batch(p, base["state"], u_ref, obs, ty, want_rows=True)
        ug, sg, rg = f.solve(tt(base["state"]), tt(u_ref), tt(obs), torch.from_numpy(ty).to(dev), want_rows=True)
        un = u_ref
    else:
        slip = np.stack([rng.uniform(-0.2, 0.2, A), rng.uniform(-0.2, 0.2, A)], axis=1)
        so_slip = slip.copy()
        uo, so, ro = oracle.solve_sa_batch(p, dt, base["state"], so_slip, u_ref, obs, ty, want_rows=True)
        sl = tt(slip)
        ug, sg, rg = f.solve(tt(base["state"]), tt(u_ref), tt(obs), torch.from_numpy(ty).to(dev), want_rows=True,
                             slip=sl, dt=dt)
    torch.cuda.synchronize()
    ug, sg, rg = ug.double().cpu().numpy(), sg.cpu().numpy(), rg.double().cpu().numpy()
    assert (sg != so).sum() <= (0 if dtype == torch.float64 else 1)
    tol = 1e-9 if dtype == torch.float64 else 1e-3
    assert (np.abs(rg - ro) / (np.abs(ro).max(axis=2, keepdims=True) + 1.0)).max() <= tol
    both = (sg == 0) & (so == 0) & (np.abs(uo[:, 1]) < 1.2)
    scale = np.maximum(1.0, np.hypot(uo[:, 0], uo[:, 1]))
    assert (np.abs(ug - uo).max(axis=1) / scale)[both].max(initial=0) <= (1e-8 if dtype == torch.float64 else 5e-3)
    if dyn == "sadbm":
        ok = (sg == so) & ((so != 0) | both)     # the slip follows the input: the agents whose inputs are compared
        assert np.abs(sl.double().cpu().numpy() - so_slip)[ok].max() <= (1e-9 if dtype == torch.float64 else 1e-4)


def test_prepared_cones_are_refused_where_they_would_silently_stop_following_the_slip_angle():
    """SADBM hands every cone the agent's slip angle after each solve (reference cbf/cbf.py:424-427); a prepared cone
    keeps the beta it was prepared with: the front ends say so instead of computing other rows than the reference's."""
    from sccav_cbf_amd import scenarios
    from sccav_cbf_amd.batched import BatchedCBF
    from sccav_cbf_amd.nominal import ClosedLoop
    dev = torch.device("cuda:0")
    w = scenarios.config4(A=64, nominal="uniform")
    obs = torch.from_numpy(w["obs"]).to(dev)
    ty = torch.from_numpy(w["obs_type"]).to(dev)
    state = torch.from_numpy(w["state"]).to(dev)
    sa = BatchedCBF(dynamics="sadbm")
    with pytest.raises(ValueError, match="frozen_cone_beta"):
        sa.prepare_obs(obs.clone(), ty.clone())
    o2, t2 = sa.prepare_obs(obs.clone(), ty.clone(), frozen_cone_beta=True)      # said so: allowed
    assert set(t2.cpu().tolist()) == {4, 5}
    with pytest.raises(ValueError, match="prepared collision cones"):
        ClosedLoop(sa, None, None, state.clone(), o2, t2, dt=0.1)
    BatchedCBF(dynamics="dbm").prepare_obs(obs.clone(), ty.clone())             # the bicycle models take them
    only_ellipses = ty.clone()
    only_ellipses[ty == 1] = 255
    sa.prepare_obs(obs.clone(), only_ellipses)                                   # no cones: nothing to refuse
