"""The nominal controller, the vehicle step and the closed loop on the GPU (through the C ABI)
against the fp64 oracle.  Needs an MI355X."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import replay  # noqa: E402
from sccav_cbf_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _course():
    cx, cy, cyaw = replay.load_course()
    return np.ascontiguousarray(np.stack([cx, cy, cyaw, np.full_like(cx, 30.0 / 3.6)], axis=1))


def _agents(A, seed, traj):
    rng = np.random.default_rng(seed)
    at = rng.integers(0, traj.shape[0], A)
    state = np.stack([traj[at, 0] + rng.normal(0, 3, A), traj[at, 1] + rng.normal(0, 3, A),
                      traj[at, 2] + rng.normal(0, 0.5, A) + 2 * np.pi * rng.integers(-2, 3, A),
                      rng.uniform(0, 15, A)], axis=1)
    last = np.where(rng.uniform(size=A) < 0.5, 0, rng.integers(0, traj.shape[0], A)).astype(np.int32)
    return np.ascontiguousarray(state), last


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("A", [1, 63, 257, 5000])
def test_nominal_matches_oracle(oracle, dtype, A):
    """LateralStanley.control + PID1.control (cbf/controllers.py:69-180) for a batch."""
    from sccav_cbf_amd.nominal import BatchedNominal
    traj = _course()
    state, last = _agents(A, 10 + A, traj)
    pid = np.random.default_rng(1).normal(0, 1, (A, 2))
    kw = dict(axle=2.9, k=0.5, ks=0.01, kp=1.2, ki=0.3, kd=0.05, dt=0.1, target_speed=8.0)
    for use_pid in (False, True):
        idx_o, pid_o = last.copy(), pid.copy()
        uo = oracle.nominal_batch(state, traj, idx_o, pid_o if use_pid else None, **kw)
        nom = BatchedNominal(**kw)
        ts = torch.from_numpy(state).to(DEV).to(dtype)
        tt = torch.from_numpy(traj).to(DEV).to(dtype)
        ti = torch.from_numpy(last.copy()).to(DEV)
        tp = torch.from_numpy(pid.copy()).to(DEV).to(dtype) if use_pid else None
        u = nom.control(ts, tt, ti, tp)
        torch.cuda.synchronize()
        u, ti = u.double().cpu().numpy(), ti.cpu().numpy()
        if dtype == torch.float64:
            assert np.array_equal(ti, idx_o)
            assert np.abs(u - uo).max() <= 1e-11
            if use_pid:
                assert np.abs(tp.cpu().numpy() - pid_o).max() <= 1e-12
        else:
            # fp32 resolves near-ties between neighbouring way-points (0.1 m apart) differently
            diff = ti != idx_o
            assert diff.mean() <= 2e-3 and (np.abs(ti - idx_o)[diff] <= 1).all()
            same = ~diff
            assert np.abs(u - uo)[same].max() <= 2e-4 * (1 + np.abs(uo).max())
            if use_pid:
                assert np.abs(tp.double().cpu().numpy() - pid_o).max() <= 1e-5 * (1 + np.abs(pid_o).max())


def test_nominal_long_course_and_waypoint_speed(oracle):
    """A course longer than one LDS chunk (2048 way-points), tie -> first index, way-point speed."""
    from sccav_cbf_amd.nominal import BatchedNominal
    N = 5000
    s = np.linspace(0, 400, N)
    traj = np.ascontiguousarray(np.stack([s, 10 * np.sin(s / 30), np.arctan(np.cos(s / 30) / 3), 5 + s / 100], axis=1))
    traj[4100] = traj[100]          # a duplicate way-point far down the course: the first one must win
    state, last = _agents(300, 3, traj)
    state[0, :3] = [traj[100, 0] - 2.9, traj[100, 1], 0.0]     # front axle exactly on the duplicate
    last[0] = 0
    idx_o = last.copy()
    uo = oracle.nominal_batch(state, traj, idx_o, None, axle=2.9, k=0.5, ks=0.0, kp=1.0)
    nom = BatchedNominal(axle=2.9, k=0.5, ks=0.0, kp=1.0)
    ts, tt = torch.from_numpy(state).to(DEV), torch.from_numpy(traj).to(DEV)
    ti = torch.from_numpy(last.copy()).to(DEV)
    u = nom.control(ts, tt, ti)
    torch.cuda.synchronize()
    assert np.array_equal(ti.cpu().numpy(), idx_o) and idx_o[0] == 100
    assert np.abs(u.cpu().numpy() - uo).max() <= 1e-11
    nom2 = BatchedNominal(axle=2.9, k=0.5, ks=0.0, kp=1.0, waypoint_speed=True)
    ti2 = torch.from_numpy(last.copy()).to(DEV)
    u2 = nom2.control(ts, tt, ti2).cpu().numpy()
    assert np.abs(u2[:, 0] - (traj[idx_o, 3] - state[:, 3])).max() <= 1e-12
    assert np.abs(u2[:, 1] - uo[:, 1]).max() <= 1e-11


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_vehicle_step_matches_oracle(oracle, dtype):
    """State.update_com, stanley_controller_ellipse.py:122-131, steering clipped at +-30 deg."""
    from sccav_cbf_amd.nominal import vehicle_step
    rng = np.random.default_rng(5)
    A = 1000
    state = np.stack([rng.uniform(-50, 50, A), rng.uniform(-50, 50, A), rng.uniform(-3.2, 3.2, A),
                      rng.uniform(0, 15, A)], axis=1)
    u = np.stack([rng.uniform(-3, 3, A), rng.uniform(-1.0, 1.0, A)], axis=1)    # beyond the clip on purpose
    so = state.copy()
    bo = oracle.vehicle_step_batch(so, u, 1.45, 1.45, np.radians(30.0), 0.1)
    ts = torch.from_numpy(state).to(DEV).to(dtype)
    tu = torch.from_numpy(u).to(DEV).to(dtype)
    tb = torch.empty((A,), dtype=dtype, device=DEV)
    vehicle_step(ts, tu, 1.45, 1.45, np.radians(30.0), 0.1, beta_out=tb)
    torch.cuda.synchronize()
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert np.abs(ts.double().cpu().numpy() - so).max() <= tol * 50
    assert np.abs(tb.double().cpu().numpy() - bo).max() <= tol
    assert np.abs(bo).max() <= np.arctan(0.5 * np.tan(np.radians(30.0))) + 1e-12


def test_small_headings_of_either_sign_keep_their_sine():
    """fp32 sine of a heading within 0.1 rad of zero, above and below: relative accuracy, read back from a vehicle
    step from the origin with unit speed, unit time step and no input (y' = sin(yaw)).  Reduced to [0, 1)
    revolutions a negative heading sat next to 1 and its sine was off by up to 14 %; a cone at the agent's own
    velocity amplifies that into its row."""
    from sccav_cbf_amd.nominal import vehicle_step
    th = 10.0 ** np.linspace(-6, -1, 2000)
    th = np.concatenate([th, -th]).astype(np.float32)
    A = th.shape[0]
    state = np.zeros((A, 4), dtype=np.float32)
    state[:, 2], state[:, 3] = th, 1.0
    ts = torch.from_numpy(state).to(DEV)
    vehicle_step(ts, torch.zeros((A, 2), dtype=torch.float32, device=DEV), 1.45, 1.45, np.radians(30.0), 1.0)
    torch.cuda.synchronize()
    got = ts.double().cpu().numpy()
    want = np.sin(th.astype(np.float64))
    assert np.abs(got[:, 1] / want - 1.0).max() <= 3e-6
    assert np.abs(got[:, 0] - np.cos(th.astype(np.float64))).max() <= 3e-7


def test_c1_closed_loop_entirely_on_the_gpu():
    """The as-shipped scenario of stanley_controller_ellipse.py (Stanley + P, one collision-cone row,
    DBM filter, update_com) with nominal, filter and vehicle step all on the device, fp64: the slip-angle
    history must reproduce the reference's beta_vs_time.mat like the CPU replay does."""
    import os
    from sccav_cbf_amd.batched import BatchedCBF, cone_slots, pack_obs
    from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
    gold = np.load(os.path.join(replay.GOLDEN, "beta_vs_time.npz"))
    traj = _course()
    k = replay.scenario_constants(traj[:, 0], traj[:, 1])
    A = 3                                                   # identical agents: rows of the batch must agree
    state = torch.tensor([[0.0, 5.0, np.radians(20.0), 10.0]] * A, dtype=torch.float64, device=DEV)
    one = torch.ones((A,), dtype=torch.float64, device=DEV)
    obs = pack_obs(cone_slots(one * k["o_cx"], one * k["o_cy"], one * 0.0, one * 0.0,
                              one * (k["a_cone"] + 1.5))[None])
    ty = torch.tensor([_lib.OBS_CONE], dtype=torch.uint8, device=DEV)
    loop = ClosedLoop(BatchedCBF(alpha=1.0, lr=replay.LR, lf=replay.LF, R=np.diag([0.5, 0.5])),
                      BatchedNominal(axle=replay.WHEELBASE, k=replay.K_STANLEY, ks=0.0, kp=replay.KP,
                                     target_speed=k["target_speed"]),
                      torch.from_numpy(traj).to(DEV), state, obs, ty, max_steer=replay.MAX_STEER, dt=replay.DT)
    # the script evaluates the target index once before the loop (stanley_controller_ellipse.py:604)
    loop.nominal.control(loop.state, loop.traj, loop.target_idx, None, u_ref=loop.u_ref)
    beta, time = [0.0], 0.0
    while k["max_time"] >= time and k["last_idx"] > int(loop.target_idx[0]):
        loop.tick()
        b = loop.beta.cpu().numpy()
        assert (b == b[0]).all() and (loop.status.cpu().numpy() == 0).all()
        beta.append(float(b[0]))
        time += replay.DT
    beta_deg = np.degrees(np.array(beta))
    ref = np.asarray(gold["beta_deg"]).reshape(-1)
    assert beta_deg.shape == ref.shape
    assert np.abs(beta_deg - ref).max() <= 1e-3
    _, beta_cpu, _ = replay.replay_stanley_cone()
    assert np.abs(beta_deg - beta_cpu).max() <= 1e-6


def _advance_planes(obs, ty, dt):
    """Constant-velocity obstacle motion on the (M,2,A,4) planes, as SCCBF_MOTION_CONSTANT_VELOCITY states it."""
    for m, kind in enumerate(ty.tolist()):
        if kind == _lib.OBS_ELLIPSE:
            obs[m, 0, :, 0] += obs[m, 1, :, 1] * dt
            obs[m, 0, :, 1] += obs[m, 1, :, 2] * dt
        elif kind == _lib.OBS_CONE:
            ang = obs[m, 0, :, 2] + obs[m, 1, :, 1]
            obs[m, 0, :, 0] += obs[m, 0, :, 3] * torch.cos(ang) * dt
            obs[m, 0, :, 1] += obs[m, 0, :, 3] * torch.sin(ang) * dt
        elif kind == _lib.OBS_RADIAL:
            obs[m, 0, :, 0] += obs[m, 1, :, 0] * dt
            obs[m, 0, :, 1] += obs[m, 1, :, 1] * dt


def _rollout_case(A, M, seed, dtype, tiled):
    from sccav_cbf_amd import scenarios
    from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled
    from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
    traj = _course()
    w = scenarios.monte_carlo(A, (M + 1) // 2, M // 2, seed, nominal="uniform")
    rng = np.random.default_rng(seed)
    at = rng.integers(0, traj.shape[0] - 300, A)
    xy = np.stack([traj[at, 0] + rng.normal(0, 2, A), traj[at, 1] + rng.normal(0, 2, A)], axis=1)
    w["obs"][:, 0, :, 0:2] += (xy - w["state"][:, 0:2])[None, :, :]
    w["state"][:, 0:2] = xy
    w["state"][:, 2] = traj[at, 2] + rng.normal(0, 0.2, A)
    w["state"][:, 3] = rng.uniform(3, 12, A)

    def make():
        state = torch.from_numpy(w["state"]).to(DEV).to(dtype)
        obs = torch.from_numpy(w["obs"]).to(DEV).to(dtype)
        ty = torch.from_numpy(w["obs_type"]).to(DEV)
        loop = ClosedLoop(BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5])),
                          BatchedNominal(axle=2.9, k=0.5, ks=0.01, kp=1.0, ki=0.2, kd=0.02, dt=0.1),
                          torch.from_numpy(traj).to(DEV).to(dtype), state,
                          planes_to_tiled(obs) if tiled else obs, ty, use_pid_state=True)
        return loop, obs, ty
    return make


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("M,motion,tiled", [(16, "constant_velocity", True), (5, "static", False),
                                           (33, "constant_velocity", False), (64, "static", True)])
def test_rollout_equals_tick_by_tick(dtype, M, motion, tiled):
    """sccbf_rollout_* (one launch) against nominal -> filter -> vehicle step -> obstacle motion
    launched tick by tick: per-tick inputs, slip angle and status, final state, statistics."""
    from sccav_cbf_amd.batched import planes_to_tiled
    A, T = 300, 25
    make = _rollout_case(A, M, 50 + M, dtype, tiled)
    la, obs_a, ty = make()
    tr = []
    hmin = torch.full((A,), float("inf"), dtype=dtype, device=DEV)
    nact = torch.zeros((A,), dtype=torch.int64, device=DEV)
    ninf = torch.zeros((A,), dtype=torch.int64, device=DEV)
    for _ in range(T):
        la.nominal.control(la.state, la.traj, la.target_idx, la.pid_state, u_ref=la.u_ref)
        u, st, rows = la.cbf.solve(la.state, la.u_ref, la.obs, ty, u_out=la.u, status=la.status, want_rows=True)
        hmin = torch.minimum(hmin, rows[:, :, 3].min(dim=0).values)
        # "the filter changed the input" = some row is violated at the nominal input (QP coordinates)
        beta_ref = torch.atan2(1.45 * torch.tan(la.u_ref[:, 1]), torch.full_like(la.u_ref[:, 1], 2.9))
        lhs = rows[:, :, 0] * la.u_ref[None, :, 0] + rows[:, :, 1] * beta_ref[None, :]
        viol = ((rows[:, :, 2] - lhs) / (torch.hypot(rows[:, :, 0], rows[:, :, 1]) + 1e-300)).max(dim=0).values
        nact += ((st == 0) & (viol > 1e-9)).long()
        ninf += (st == 1).long()
        from sccav_cbf_amd.nominal import vehicle_step
        vehicle_step(la.state, la.u, lr=1.45, lf=1.45, max_steer=la.max_steer, dt=la.dt, beta_out=la.beta)
        tr.append(torch.cat([u, la.beta[:, None], st.to(dtype)[:, None]], dim=1).clone())
        if motion == "constant_velocity":
            _advance_planes(obs_a, ty, la.dt)
            la.obs = planes_to_tiled(obs_a) if tiled else obs_a
    tr = torch.stack(tr)
    lb, _, _ = make()
    out = lb.rollout(T, motion=motion, trace=True)
    torch.cuda.synchronize()
    # fp32: the two paths round differently (the compiler fuses multiply-adds per inlined copy) and 25 closed-loop ticks
    # amplify last-bit differences: a detector of gross errors, as in tests/campaigns/fuzz_rollout.py (5e-2 there)
    tol = 1e-7 if dtype == torch.float64 else 2e-2
    same = (out["trace"][:, :, 3] == tr[:, :, 3]).all(dim=0)       # agents whose status history agrees
    assert same.float().mean().item() >= (1.0 if dtype == torch.float64 else 0.97)
    # fp32: an agent whose status differed on one tick (boundary of feasibility) took another path
    # (a vertex on two nearly parallel rows gives inputs of 1e4 and more: errors scale with them)
    scale = 1 + tr[:, :, :2].abs().amax(dim=(0, 2))
    # An agent whose optimum once sat on two nearly parallel rows was handed an input of 1e4 m/s^2 and more and leaves
    # every physical regime: from then on its loop multiplies a last-bit difference between the two paths tick by tick
    # (tools/diag_rollout_diff.py prints such a history).  Those few are held to a gross-error bar only.
    thrown = scale > 1e3
    assert thrown.float().mean().item() <= 0.15
    dd = (out["trace"][:, :, :3] - tr[:, :, :3]).abs().amax(dim=(0, 2)) / scale
    d = dd[same & ~thrown].max().item()
    assert d <= tol, d
    assert dd[same].max().item() <= max(tol, 1e-4), dd[same].max().item()
    assert ((lb.state - la.state).abs().amax(dim=1) / scale)[same & ~thrown].max().item() <= tol * 50
    di = (lb.target_idx[same & ~thrown] - la.target_idx[same & ~thrown]).abs()
    if dtype == torch.float64:
        assert di.max().item() == 0
    else:   # near-ties between way-points 0.1 m apart
        assert di.max().item() <= 2 and (di == 0).float().mean().item() >= 0.97
    assert ((lb.pid_state - la.pid_state).abs().amax(dim=1) / scale)[same & ~thrown].max().item() <= tol * 50
    assert (out["h_min"][same] - hmin[same]).abs().max().item() <= tol * (1 + hmin[same].abs().max().item())
    assert torch.equal(out["n_infeasible"][same].long(), ninf[same])
    if dtype == torch.float64:
        dn = (out["n_active"].long() - nact).abs()
        assert dn.max().item() <= 1 and (dn == 0).float().mean().item() >= 0.99


def test_c1_by_one_rollout_launch():
    """The as-shipped scenario in ONE launch of 276 ticks (fp64): the slip-angle trace reproduces the
    reference's beta_vs_time.mat."""
    import os
    from sccav_cbf_amd.batched import BatchedCBF, cone_slots, pack_obs
    from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
    gold = np.load(os.path.join(replay.GOLDEN, "beta_vs_time.npz"))
    ref = np.asarray(gold["beta_deg"]).reshape(-1)
    traj = _course()
    k = replay.scenario_constants(traj[:, 0], traj[:, 1])
    A = 2
    state = torch.tensor([[0.0, 5.0, np.radians(20.0), 10.0]] * A, dtype=torch.float64, device=DEV)
    one = torch.ones((A,), dtype=torch.float64, device=DEV)
    obs = pack_obs(cone_slots(one * k["o_cx"], one * k["o_cy"], one * 0.0, one * 0.0,
                              one * (k["a_cone"] + 1.5))[None])
    ty = torch.tensor([_lib.OBS_CONE], dtype=torch.uint8, device=DEV)
    loop = ClosedLoop(BatchedCBF(alpha=1.0, lr=replay.LR, lf=replay.LF, R=np.diag([0.5, 0.5])),
                      BatchedNominal(axle=replay.WHEELBASE, k=replay.K_STANLEY, ks=0.0, kp=replay.KP,
                                     target_speed=k["target_speed"]),
                      torch.from_numpy(traj).to(DEV), state, obs, ty, max_steer=replay.MAX_STEER, dt=replay.DT)
    out = loop.rollout(len(ref) - 1, trace=True)
    torch.cuda.synchronize()
    beta_deg = np.concatenate([[0.0], np.degrees(out["trace"][:, 0, 2].cpu().numpy())])
    assert np.abs(beta_deg - ref).max() <= 1e-3
    assert int(out["n_active"][0]) == 59 and int(out["n_infeasible"][0]) == 0      # ticks with an active row
    assert float(out["h_min"][0]) > -1e-6                                          # the barrier is never crossed


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_rollout_seekers_equal_tick_by_tick(oracle, dtype):
    """SCCBF_MOTION_SEEKER inside the launch against nominal -> filter -> vehicle step launched tick
    by tick with the oracle's seeker step (radial_dynamic_obstacles.py:193-239) moving the obstacles
    on the host between ticks: ellipse and radial slots seek, cones drift."""
    from sccav_cbf_amd.nominal import vehicle_step
    A, M, T = 200, 12, 20
    make = _rollout_case(A, M, 77, dtype, False)
    la, obs_a, ty = make()
    ty_np = ty.cpu().numpy().copy()
    ty_np[0:3] = _lib.OBS_RADIAL          # ellipse slots read as radial ones: cx, cy, a, b, cdot, kv
    obs_np = obs_a.double().cpu().numpy().copy()
    obs_np[0:3, 1, :, 2] = 1.0            # kv
    ty = torch.from_numpy(ty_np).to(DEV)

    def fresh():
        loop, _, _ = make()
        loop.obs = torch.from_numpy(obs_np).to(DEV).to(dtype)
        loop.obs_type = ty
        return loop
    la = fresh()
    host = obs_np.copy()
    tr = []
    for _ in range(T):
        la.nominal.control(la.state, la.traj, la.target_idx, la.pid_state, u_ref=la.u_ref)
        u, st = la.cbf.solve(la.state, la.u_ref, la.obs, ty, u_out=la.u, status=la.status)
        vehicle_step(la.state, la.u, lr=1.45, lf=1.45, max_steer=la.max_steer, dt=la.dt, beta_out=la.beta)
        tr.append(torch.cat([u, la.beta[:, None], st.to(dtype)[:, None]], dim=1).clone())
        oracle.seeker_step(host, ty_np, la.state[:, 0:2].double().cpu().numpy(), la.dt, k=0.3, v_min=2.0)
        la.obs = torch.from_numpy(host).to(DEV).to(dtype)
    tr = torch.stack(tr)
    lb = fresh()
    out = lb.rollout(T, motion="seeker", trace=True, seek_gain=0.3, seek_v_min=2.0)
    torch.cuda.synchronize()
    # 20 closed-loop ticks among crowding seekers multiply the last-bit differences between the two
    # kernels (same formulas, compiled into different code: other fused multiply-adds) a hundredfold
    tol = 1e-6 if dtype == torch.float64 else 5e-3
    same = (out["trace"][:, :, 3] == tr[:, :, 3]).all(dim=0)
    # fp64: seekers crowd the agents, and one whose rows close to a sliver is driven to |a| ~ 1e5
    # and speeds ~ 1e5 m/s, where the feasibility decision is a coin toss between the two runs
    assert same.float().mean().item() >= (0.99 if dtype == torch.float64 else 0.97)
    scale = 1 + tr[:, :, :2].abs().amax(dim=(0, 2))
    # such an agent (inputs beyond 1e3 m/s^2) doubles any last-bit difference on every tick: left out
    same &= scale < 1e3
    assert same.float().mean().item() >= 0.95
    d = ((out["trace"][:, :, :3] - tr[:, :, :3]).abs().amax(dim=(0, 2)) / scale)[same].max().item()
    assert d <= tol, d
    assert ((lb.state - la.state).abs().amax(dim=1) / scale)[same].max().item() <= tol * 50


def test_radial_seeker_scenario_by_one_rollout_launch(oracle):
    """The radial scenario (ego at rest, nominal input (0, 0), 32 circular seekers closing in;
    radial_dynamic_obstacles.py:427-487) for 16 egos in ONE launch without a course, fp64, against
    the same loop done entirely by the oracle on the host."""
    from sccav_cbf_amd import scenarios
    from sccav_cbf_amd.batched import BatchedCBF
    from sccav_cbf_amd.nominal import ClosedLoop
    A, M, T, dt = 16, 32, 90, 1.0 / 30.0
    w = scenarios.radial_seekers(M=M, A=A, seed=1236)
    state = torch.from_numpy(w["state"]).to(DEV)
    loop = ClosedLoop(BatchedCBF(alpha=1.0), None, None, state, torch.from_numpy(w["obs"]).to(DEV),
                      torch.from_numpy(w["obs_type"]).to(DEV), dt=dt)
    out = loop.rollout(T, motion="seeker", trace=True)
    torch.cuda.synchronize()
    prm = oracle.make_params(alpha=1.0)
    s, obs = w["state"].copy(), w["obs"].copy()
    zero = np.zeros((A, 2))
    tr = np.empty((T, A, 4))
    hmin = np.full(A, np.inf)
    for t in range(T):
        u, st, rows = oracle.solve_batch(prm, s, zero, obs, w["obs_type"], want_rows=True)
        hmin = np.minimum(hmin, rows[:, :, 3].min(axis=0))
        beta = oracle.vehicle_step_batch(s, u, 1.45, 1.45, np.radians(30.0), dt)
        tr[t, :, 0:2], tr[t, :, 2], tr[t, :, 3] = u, beta, st
        oracle.seeker_step(obs, w["obs_type"], s[:, 0:2], dt)
    got = out["trace"].cpu().numpy()
    assert np.array_equal(got[:, :, 3], tr[:, :, 3])
    assert (tr[:, :, 3] == 0).any() and (np.abs(tr[:, :, 0]) > 1e-3).any()       # the filter did act
    assert np.abs(got[:, :, :3] - tr[:, :, :3]).max() <= 1e-7 * (1 + np.abs(tr[:, :, :3]).max())
    assert np.abs(loop.state.cpu().numpy() - s).max() <= 1e-7 * (1 + np.abs(s).max())
    assert np.abs(out["h_min"].cpu().numpy() - hmin).max() <= 1e-7 * (1 + np.abs(hmin).max())


def test_c1_with_the_true_ellipse_row_by_one_rollout_launch():
    """The same scenario constrained by the ellipse itself (Ellipse2D a = 20, b = 10 at the obstacle's
    centre; the inline CBF_A of stanley_controller_ellipse.py:277-318) in one launch, against the
    oracle's replay of that loop tick by tick; the ego never enters the ellipse."""
    from sccav_cbf_amd.batched import BatchedCBF, ellipse_slots, pack_obs
    from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
    t_ref, beta_ref, log = replay.replay_stanley_cone(row_kind="ellipse")
    traj = _course()
    k = replay.scenario_constants(traj[:, 0], traj[:, 1])
    A = 3
    state = torch.tensor([[0.0, 5.0, np.radians(20.0), 10.0]] * A, dtype=torch.float64, device=DEV)
    one = torch.ones((A,), dtype=torch.float64, device=DEV)
    zero = one * 0.0
    obs = pack_obs(ellipse_slots(one * k["o_cx"], one * k["o_cy"], one * k["a"], one * k["b"], zero, zero, zero)[None])
    ty = torch.tensor([_lib.OBS_ELLIPSE], dtype=torch.uint8, device=DEV)
    loop = ClosedLoop(BatchedCBF(alpha=1.0, lr=replay.LR, lf=replay.LF, R=np.diag([0.5, 0.5])),
                      BatchedNominal(axle=replay.WHEELBASE, k=replay.K_STANLEY, ks=0.0, kp=replay.KP,
                                     target_speed=k["target_speed"]),
                      torch.from_numpy(traj).to(DEV), state, obs, ty, max_steer=replay.MAX_STEER, dt=replay.DT)
    T = len(t_ref) - 1
    out = loop.rollout(T, trace=True)
    torch.cuda.synchronize()
    tr = out["trace"][:, 0, :].cpu().numpy()
    assert np.abs(np.degrees(tr[:, 2]) - beta_ref[1:]).max() <= 1e-6
    assert np.abs(tr[:, 0] - np.array([l[3] for l in log])).max() <= 1e-7
    assert (tr[:, 3] == 0).all() and float(out["h_min"][0]) > -1e-6 and int(out["n_active"][0]) > 10


def test_target_index_beyond_the_course_reads_the_last_way_point(oracle):
    """A caller's stale index (a longer course before) must not read past the course."""
    traj = _course()[:200]
    A = 300
    state, _ = _agents(A, 21, traj)
    tgt = np.random.default_rng(1).integers(0, 5000, A).astype(np.int32)
    tgt_o = tgt.copy()
    uo = oracle.nominal_batch(state, traj, tgt_o, axle=2.9, k=0.5, ks=0.01)
    from sccav_cbf_amd.nominal import BatchedNominal
    tg = torch.from_numpy(tgt.copy()).to(DEV)
    ug = BatchedNominal(axle=2.9, k=0.5, ks=0.01).control(torch.from_numpy(state).to(DEV), torch.from_numpy(traj).to(DEV), tg)
    torch.cuda.synchronize()
    assert np.array_equal(tg.cpu().numpy(), tgt_o)
    assert np.abs(ug.cpu().numpy() - uo).max() <= 1e-9


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_nonfinite_and_diverged_states_do_not_disturb_their_neighbours(dtype):
    """NaN, infinite and absurdly large states in a batch: the kernels return, the other agents'
    results are what they are without those agents, and the rollout flags the bad ones."""
    from sccav_cbf_amd.nominal import BatchedNominal
    A, M, T = 300, 8, 6
    make = _rollout_case(A, M, 91, dtype, True)
    clean, _, _ = make()
    out_clean = clean.rollout(T, trace=True)
    dirty, _, _ = make()
    bad = torch.tensor([3, 64, 65, 130, 299], device=DEV)
    vals = torch.tensor([[float("nan"), 0, 0, 5], [0, float("inf"), 0, 5], [1e30, 1e30, 1e17, 5],
                         [0, 0, float("-inf"), 5], [0, 0, 0, float("nan")]], dtype=dtype, device=DEV)
    dirty.state[bad] = vals
    state_in = dirty.state.clone()
    tgt = torch.zeros((A,), dtype=torch.int32, device=DEV)
    u = BatchedNominal(axle=2.9).control(state_in, dirty.traj, tgt)
    torch.cuda.synchronize()
    good = torch.ones(A, dtype=torch.bool, device=DEV)
    good[bad] = False
    assert torch.isfinite(u[good]).all() and (tgt >= 0).all() and (tgt < dirty.traj.shape[0]).all()
    out = dirty.rollout(T, trace=True)
    torch.cuda.synchronize()
    assert torch.equal(out["trace"][:, good], out_clean["trace"][:, good])
    assert torch.equal(dirty.state[good], clean.state[good])
    st0 = out["trace"][0, bad, 3]
    assert (st0[[0, 1, 3, 4]] == 2).all()          # non-finite state: flagged on the first tick


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("lanes,M,tiled", [(0, 16, True), (4, 16, False), (16, 5, True), (64, 16, False), (0, 300, False), (0, 0, False)])
def test_solve_step_is_solve_then_vehicle_step(dtype, lanes, M, tiled):
    """The fused tick (filter + vehicle step in one launch) against the two launches: same inputs and
    verdicts bit for bit, the stepped state and slip angle to rounding (the step is compiled into a
    different kernel)."""
    from sccav_cbf_amd import scenarios
    from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled
    from sccav_cbf_amd.nominal import vehicle_step
    A = 3000
    w = scenarios.config4(A=A, seed=31)
    reps = -(-max(M, 1) // 16)
    obs = torch.from_numpy(np.tile(w["obs"], (reps, 1, 1, 1))[:M].copy()).to(dtype).to(DEV)
    ty = torch.from_numpy(np.tile(w["obs_type"], reps)[:M].copy()).to(DEV)
    if tiled:
        obs = planes_to_tiled(obs)
    state = torch.from_numpy(w["state"]).to(dtype).to(DEV)
    u_ref = torch.from_numpy(w["u_ref"]).to(dtype).to(DEV)
    f = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), lanes_per_qp=lanes)
    u0, st0 = f.solve(state, u_ref, obs, ty)
    s0 = state.clone()
    b0 = torch.empty(A, dtype=dtype, device=DEV)
    vehicle_step(s0, u0, max_steer=np.radians(30.0), dt=0.1, beta_out=b0)
    s1 = state.clone()
    b1 = torch.empty(A, dtype=dtype, device=DEV)
    u1, st1 = f.solve(s1, u_ref, obs, ty, step_dt=0.1, max_steer=np.radians(30.0), beta_out=b1)
    torch.cuda.synchronize()
    assert torch.equal(st0, st1)
    assert torch.equal(u0, u1)
    fin = torch.isfinite(s0).all(dim=1)
    assert torch.equal(fin, torch.isfinite(s1).all(dim=1))
    tol = 1e-13 if dtype == torch.float64 else 1e-6
    scale = 1 + s0[fin].abs().amax(dim=1, keepdim=True)
    assert ((s1[fin] - s0[fin]).abs() / scale).max() <= tol
    assert (b1[fin] - b0[fin]).abs().max() <= tol
    assert not torch.equal(s1, state)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_nearest_waypoint_of_agents_far_off_the_course(dtype):
    """Agents thrown 1e3 ... 1e12 m off the course: in the working precision more and more way-points
    lie at the same computed distance, down to all of them.  The answer is still the first index of
    the least computed squared distance fma(ex, ex, ey * ey), and the search must not fall back to
    scanning the whole course for it (that is a matter of time, measured elsewhere; here: the index)."""
    from sccav_cbf_amd.nominal import BatchedNominal
    npdt = np.float64 if dtype == torch.float64 else np.float32
    traj = _course().astype(npdt)
    N = traj.shape[0]
    rng = np.random.default_rng(11)
    A = 4096
    mag = 10.0 ** rng.uniform(3, 12, A)
    ang = rng.uniform(0, 2 * np.pi, A)
    state = np.zeros((A, 4), dtype=npdt)
    state[:, 0] = (50 + mag * np.cos(ang)).astype(npdt)
    state[:, 1] = (-10 + mag * np.sin(ang)).astype(npdt)
    state[:, 3] = 5.0
    last = rng.integers(0, N, A).astype(np.int32)
    last[::3] = 0
    nom = BatchedNominal(axle=0.0)          # front axle = reference point: fx = x exactly
    tgt = torch.from_numpy(last.copy()).to(DEV)
    nom.control(torch.from_numpy(state).to(DEV), torch.from_numpy(traj).to(DEV), tgt)
    torch.cuda.synchronize()
    got = tgt.cpu().numpy()
    # the same roundings on the host: ex, ey in the working precision, ey * ey rounded, then one fused sum
    ex = (state[:, None, 0] - traj[None, :, 0]).astype(npdt).astype(np.float64)
    ey = (state[:, None, 1] - traj[None, :, 1]).astype(npdt).astype(np.float64)
    ey2 = (ey * ey).astype(npdt).astype(np.float64)
    if dtype == torch.float32:
        d = (ex * ex + ey2).astype(npdt)            # 48 + 24 bits: exact in float64 but for rare double roundings
    else:
        import math
        d = np.empty_like(ex)
        fma = getattr(math, "fma", None)
        for i in range(A):                          # float64: no wider type to hide the fused rounding in
            d[i] = [fma(a, a, b) for a, b in zip(ex[i], ey2[i])] if fma else ex[i] * ex[i] + ey2[i]
    nearest = d.argmin(axis=1)                      # first index of the least value
    want = np.maximum(last, nearest)                # the target index never moves backwards
    ok = got == want
    tie = d[np.arange(A), np.minimum(got, N - 1)] == d.min(axis=1)
    if dtype == torch.float32:
        assert ok.mean() >= 0.999 and (ok | tie).all()
    else:
        # Python 3.10 has no math.fma: the unfused sum may differ in the last bit, which can reorder exact-looking ties
        assert ok.mean() >= 0.98 and (ok | (np.abs(d[np.arange(A), np.minimum(got, N - 1)] - d.min(axis=1))
                                            <= 4e-16 * d.min(axis=1))).all()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("mode", ["acc", "vel", "unicycle"])
def test_kinematic_steps_match_the_restated_driver_steps(oracle, dtype, mode):
    from sccav_cbf_amd.nominal import vehicle_step_kin
    rng = np.random.default_rng(5)
    A = 3000
    state = np.stack([rng.uniform(-50, 50, A), rng.uniform(-50, 50, A), rng.uniform(-7, 7, A), rng.uniform(0, 15, A)], 1)
    u = np.stack([rng.uniform(-3, 12, A), rng.uniform(-0.9, 0.9, A)], 1)
    s = torch.from_numpy(state).to(DEV).to(dtype)
    want = s.double().cpu().numpy().copy()
    ud = torch.from_numpy(u).to(DEV).to(dtype)
    vehicle_step_kin(s, ud, mode, L=2.9, max_steer=np.radians(30.0), dt=0.1)
    oracle.vehicle_step_kin_batch(want, ud.double().cpu().numpy(), mode, 2.9, np.radians(30.0), 0.1)
    torch.cuda.synchronize()
    got = s.double().cpu().numpy()
    d = np.abs(got - want)
    d[:, 2] = np.minimum(d[:, 2], np.abs(d[:, 2] - 2 * np.pi))     # -pi and pi are the same heading
    assert d.max() <= (1e-12 if dtype == torch.float64 else 2e-5)


def _model_loop(dyn, A, M, seed, dtype, N=None, tiled=False):
    from sccav_cbf_amd import scenarios
    from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled
    from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
    traj = _course()
    if N is not None:          # a longer course: the same curve sampled more finely
        t = np.linspace(0, traj.shape[0] - 1, N)
        traj = np.stack([np.interp(t, np.arange(traj.shape[0]), traj[:, k]) for k in range(4)], axis=1)
    w = scenarios.monte_carlo(A, (M + 1) // 2, M // 2, seed, nominal="uniform")
    rng = np.random.default_rng(seed)
    at = rng.integers(0, traj.shape[0] - traj.shape[0] // 6, A)
    xy = np.stack([traj[at, 0] + rng.normal(0, 2, A), traj[at, 1] + rng.normal(0, 2, A)], axis=1)
    w["obs"][:, 0, :, 0:2] += (xy - w["state"][:, 0:2])[None, :, :]
    w["state"][:, 0:2] = xy
    w["state"][:, 2] = traj[at, 2] + rng.normal(0, 0.2, A)
    w["state"][:, 3] = rng.uniform(3, 12, A)

    def make():
        obs = torch.from_numpy(w["obs"]).to(DEV).to(dtype)
        return ClosedLoop(BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), dynamics=dyn),
                          BatchedNominal(axle=2.9, k=0.5, ks=0.01, kp=1.0, ki=0.2, kd=0.02, dt=0.1),
                          torch.from_numpy(traj).to(DEV).to(dtype), torch.from_numpy(w["state"]).to(DEV).to(dtype),
                          planes_to_tiled(obs) if tiled else obs, torch.from_numpy(w["obs_type"]).to(DEV),
                          use_pid_state=True)
    return make


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn,M,tiled", [("kbm", 12, False), ("dum", 16, True), ("sadbm", 9, False), ("sadbm", 40, True),
                                         ("dbm", 100, True), ("dbm", 300, False), ("sadbm", 130, False),
                                         ("dum", 1000, True)])   # beyond 64 slots: one wavefront per agent
def test_rollout_of_every_model_equals_its_ticks(dtype, dyn, M, tiled):
    """sccbf_rollout_* for the kinematic, unicycle-input and steer-rate models against ClosedLoop.tick
    (nominal -> the model's solve entry -> the model's vehicle step) launched tick by tick."""
    A, T = (300, 20) if M <= 64 else (40, 8)
    make = _model_loop(dyn, A, M, 90 + M, dtype, tiled=tiled)
    la = make()
    tr = []
    for _ in range(T):
        la.tick()
        tr.append(torch.cat([la.u, la.status.to(dtype)[:, None]], dim=1).clone())
    tr = torch.stack(tr)
    lb = make()
    out = lb.rollout(T, trace=True)
    torch.cuda.synchronize()
    tol = 1e-6 if dtype == torch.float64 else 5e-3
    same = (out["trace"][:, :, 3] == tr[:, :, 2]).all(dim=0)
    assert same.float().mean().item() >= (0.99 if dtype == torch.float64 else 0.95)
    scale = 1 + tr[:, :, :2].abs().amax(dim=(0, 2))
    same &= scale < 1e3                               # agents driven off by an unbounded QP amplify every last bit
    assert same.float().mean().item() >= 0.8
    d = ((out["trace"][:, :, :2] - tr[:, :, :2]).abs().amax(dim=(0, 2)) / scale)[same].max().item()
    assert d <= tol, d
    ds = (lb.state - la.state).abs()
    ds[:, 2] = torch.minimum(ds[:, 2], (ds[:, 2] - 2 * np.pi).abs())
    assert (ds.amax(dim=1) / scale)[same].max().item() <= tol * 50
    assert (lb.target_idx[same] - la.target_idx[same]).abs().max().item() <= (0 if dtype == torch.float64 else 1)
    if dyn == "sadbm":
        assert ((lb.slip - la.slip).abs().amax(dim=1) / scale)[same].max().item() <= tol * 50


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_rollout_on_a_course_longer_than_one_staging_pass(dtype):
    """5000 way-points: the rollout stages the course pass by pass on every tick; against the ticks."""
    A, T = 200, 12
    make = _model_loop("dbm", A, 8, 7, dtype, N=5000)
    la = make()
    tr = []
    for _ in range(T):
        la.tick()
        tr.append(torch.cat([la.u, la.status.to(dtype)[:, None]], dim=1).clone())
    tr = torch.stack(tr)
    lb = make()
    out = lb.rollout(T, trace=True)
    torch.cuda.synchronize()
    tol = 1e-6 if dtype == torch.float64 else 5e-3
    same = (out["trace"][:, :, 3] == tr[:, :, 2]).all(dim=0)
    assert same.float().mean().item() >= (0.99 if dtype == torch.float64 else 0.95)
    scale = 1 + tr[:, :, :2].abs().amax(dim=(0, 2))
    same &= scale < 1e3
    assert same.float().mean().item() >= 0.8
    assert ((out["trace"][:, :, :2] - tr[:, :, :2]).abs().amax(dim=(0, 2)) / scale)[same].max().item() <= tol
    assert (lb.target_idx[same] - la.target_idx[same]).abs().max().item() <= (0 if dtype == torch.float64 else 1)
