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
