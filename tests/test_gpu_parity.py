"""Parity of the HIP path (through the C ABI) with the fp64 oracle.  Needs an MI355X.

Bars
  fp64   status identical; |du| <= 1e-8 max(1, |u|)
  fp32   stated fp64 -> fp32 tolerance: feasibility a.u - b >= -3e-4 (|a||u| + |b|);
         cost within 2e-3 (1 + cost); |du| <= 1e-3 max(1, |u|) where the active pair is
         well conditioned (|sin| >= 1e-3); status identical except for at most 1 agent in
         4000 whose rows are within 1e-3 (relative) of the feasibility boundary
"""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sccav_cbf_amd import _lib, scenarios  # noqa: E402
from tests import parity  # noqa: E402

pytestmark = pytest.mark.gpu

R_HALF = np.diag([0.5, 0.5])
ALL_LANES = (0, 1, 2, 4, 8, 16, 32, 64)


def _dev():
    return torch.device("cuda:0")


def _filter(lanes=0, slip_io=False, alpha=1.0, R=R_HALF, dynamics="dbm", **kw):
    from sccav_cbf_amd.batched import BatchedCBF
    return BatchedCBF(alpha=alpha, R=R, lanes_per_qp=lanes, slip_io=slip_io, dynamics=dynamics, **kw)


def _to_dev(w, dtype):
    t = {}
    for k, v in w.items():
        x = torch.from_numpy(np.ascontiguousarray(v)).to(_dev())
        t[k] = x.to(dtype).contiguous() if x.dtype.is_floating_point else x
    return t


def _gpu_solve(w, dtype, lanes, slip_io=True, want_rows=True, **kw):
    f = _filter(lanes=lanes, slip_io=slip_io, **kw)
    t = _to_dev(w, dtype)
    out = f.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], want_rows=want_rows)
    torch.cuda.synchronize()
    return [o.double().cpu().numpy() if o.dtype.is_floating_point else o.cpu().numpy() for o in out]


def _orc_solve(oracle, w, slip_io=True, alpha=1.0, R=R_HALF, dynamics=0, extra_flags=0):
    flags = (oracle.FLAG_SLIP_IO if slip_io else 0) | extra_flags
    p = oracle.make_params(alpha=alpha, R=np.asarray(R).reshape(-1), flags=flags, dynamics=dynamics)
    return oracle.solve_batch(p, w["state"], w["u_ref"], w["obs"], w["obs_type"], want_rows=True)


def _near_feasibility_boundary(idx, so, ro, un, R, rel=1e-3):
    """True when moving every right-hand side by `rel` of its row's scale flips the oracle's
    feasibility verdict for each listed agent."""
    from oracle import oracle as orc
    rows = np.ascontiguousarray(ro[:, idx, :])
    scale = np.hypot(rows[:, :, 0], rows[:, :, 1]) * (1 + np.hypot(un[idx, 0], un[idx, 1]))[None, :] \
        + np.abs(rows[:, :, 2])
    p = orc.make_params(R=np.asarray(R).reshape(-1))
    for k, a in enumerate(idx):
        sign = 1.0 if so[a] == 0 else -1.0          # tighten an optimal one, loosen an infeasible one
        r = rows[:, k:k + 1, :].copy()
        r[:, 0, 2] += sign * rel * scale[:, k]
        _, st = orc.qp_batch(p, r, un[a:a + 1])
        if st[0] == so[a]:
            return False
    return True


def _check(ug, sg, uo, so, ro, un, dtype, R=R_HALF):
    mism = np.where(sg != so)[0]
    if dtype == torch.float64:
        assert mism.size == 0, "status mismatch on %d agents" % mism.size
    else:
        # fp32 may decide differently only within 1e-3 of the feasibility boundary
        assert mism.size <= max(1, len(so) // 4000), "status mismatch on %d agents" % mism.size
        assert ((sg[mism] == 2) == (so[mism] == 2)).all()
        # ... or where fp64 finds an optimum on two rows that fp32 cannot tell from parallel
        # (the kernel treats |sin| < 1e-6 as parallel; such optima lie ~1e5 away from u_ref)
        det = parity.active_det(uo, ro)
        wedge = (so[mism] == 0) & (det[mism] < 1e-5)
        assert _near_feasibility_boundary(mism[~wedge], so, ro, un, R)
    both = (so == 0) & (sg == 0)
    scale = np.maximum(1.0, np.hypot(uo[:, 0], uo[:, 1]))
    du = np.abs(ug - uo).max(axis=1) / scale
    if dtype == torch.float64:
        assert du[both].max(initial=0) <= 1e-8, du[both].max()
        assert np.array_equal(ug[~both], uo[~both])       # pass-through is exact
        return
    cost_o, _ = parity.qp_metrics(uo, un, ro, R)
    cost_g, viol_g = parity.qp_metrics(ug, un, ro, R)
    assert viol_g[both].max(initial=0) <= 3e-4, viol_g[both].max()
    # a vertex on two nearly parallel rows moves by (row error) / |sin|: cost and u are
    # compared where |sin| >= 1e-3, feasibility everywhere
    well = both & (parity.active_det(uo, ro) >= 1e-3)
    assert well.sum() >= 0.95 * both.sum()      # lane + cone rows: ~3 % ill-conditioned pairs
    assert (np.abs(cost_g - cost_o)[well] / (1 + cost_o[well])).max(initial=0) <= 2e-3
    assert du[well].max(initial=0) <= 1e-3, du[well].max()


SCENARIOS = {
    "mixed16": lambda: scenarios.config4(A=4096, nominal="uniform"),
    "ellipse8": lambda: scenarios.config3(A=4096),
    "lanes2_cones6": lambda: scenarios.lanes_and_cones(A=4096),
    "radial32": lambda: scenarios.radial_seekers(M=32, A=257, seed=5),
    # lane polynomials of degree 5 and 6 (kind 8: seven coefficients and the buffer)
    "lanes6_cones6": lambda: scenarios.lanes_and_cones(A=4096, curved=True),
    # the same obstacles with their per-obstacle constants evaluated into the slot (kinds 4, 5)
    "mixed16_prepared": lambda: scenarios.prepared(scenarios.config4(A=4096, nominal="uniform")),
    "lanes2_cones6_prepared": lambda: scenarios.prepared(scenarios.lanes_and_cones(A=4096)),
}


@pytest.mark.parametrize("name", list(SCENARIOS))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("lanes", ALL_LANES)
def test_solve_matches_oracle(oracle, name, dtype, lanes):
    w = SCENARIOS[name]()
    M = w["obs"].shape[0]
    if lanes not in (0, 32, 64) and M > 16 * lanes:
        pytest.skip("this variant holds at most 16 rows per lane")
    uo, so, ro = _orc_solve(oracle, w)
    ug, sg, rg = _gpu_solve(w, dtype, lanes)
    _check(ug, sg, uo, so, ro, w["u_ref"], dtype)
    # rows: every element against the magnitude of its row (h and b come out of cancelling terms as
    # large as the row's normal, so their error scales with the row, not with themselves)
    tol = 1e-9 if dtype == torch.float64 else 1e-3
    assert (np.abs(rg - ro) / (np.abs(ro).max(axis=2, keepdims=True) + 1.0)).max() <= tol


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_steering_transform_path(oracle, dtype):
    """delta -> beta on the way in, beta -> delta on the way out (cbf/cbf.py:175, 216)."""
    w = scenarios.config3(A=2048)
    uo, so, _ = _orc_solve(oracle, w, slip_io=False)
    ug, sg = _gpu_solve(w, dtype, 0, slip_io=False, want_rows=False)
    assert (sg == so).all()
    ok = (so == 0) & (np.abs(uo[:, 1]) < 1.2)      # tan is ill-conditioned near pi/2
    tol = 1e-8 if dtype == torch.float64 else 2e-3
    assert np.abs(ug - uo)[ok].max() <= tol
    assert np.allclose(ug[so != 0], w["u_ref"][so != 0], rtol=0, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_kbm_dynamics(oracle, dtype):
    """KBM_VC_CBF2D rows and steering maps (cbf/cbf.py:75, 94-101, 109)."""
    w = scenarios.config3(A=2048)
    w["u_ref"][:, 0] = np.abs(w["u_ref"][:, 0]) + 2.0        # v_ref > 0
    w["u_ref"][:, 1] *= 0.5
    for slip_io in (True, False):
        uo, so, ro = _orc_solve(oracle, w, slip_io=slip_io, dynamics=1, R=np.eye(2))
        ug, sg, rg = _gpu_solve(w, dtype, 0, slip_io=slip_io, dynamics="kbm", R=np.eye(2))
        assert (sg == so).all()
        ok = so == 0
        tol = 1e-8 if dtype == torch.float64 else 2e-3
        assert (np.abs(ug - uo)[ok].max(axis=1) / np.maximum(1, np.abs(uo[ok]).max(axis=1))).max() <= tol


def test_partials_match_oracle(oracle):
    w = scenarios.lanes_and_cones(A=64, n_cone=3)
    we = scenarios.monte_carlo(64, 3, 0, 9, nominal="uniform")
    wr = scenarios.radial_seekers(M=2, A=64, seed=3)
    wr["state"] = we["state"] * 0.1
    lib = _lib.load()
    for ww in (w, we, wr):
        for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 2e-3)):
            t = _to_dev(ww, dtype)
            M, A = ww["obs"].shape[0], ww["state"].shape[0]
            out = torch.empty((M, 2, A, 4), dtype=dtype, device=_dev())
            p = _lib.Params()
            lib.sccbf_default_params(C.byref(p))
            fn = lib.sccbf_partials_f64 if dtype == torch.float64 else lib.sccbf_partials_f32
            _lib.check(fn(C.byref(p), A, M, C.c_void_p(t["state"].data_ptr()), C.c_void_p(t["obs"].data_ptr()),
                          C.c_void_p(t["obs_type"].data_ptr()), C.c_void_p(out.data_ptr()), None))
            torch.cuda.synchronize()
            o = out.double().cpu().numpy()
            for m in range(M):
                for a in range(0, A, 7):
                    slot = ww["obs"][m, :, a, :].reshape(8)
                    d = oracle.barrier(int(ww["obs_type"][m]), ww["state"][a], slot)
                    got = np.concatenate([o[m, 0, a, :], o[m, 1, a, :2]])
                    exp = np.array([d["h"], d["hx"], d["hy"], d["hth"], d["hv"], d["ht"]])
                    assert np.abs(got - exp).max() <= tol * (1 + np.abs(exp).max()), (m, a, got, exp)


def _adversarial_rows():
    cases = [
        [(1, 0, 1.0), (1, 0, 1.0), (2, 0, 2.0), (1, 1e-13, 1.0), (0, 1, -3)],     # duplicates / parallel
        [(1, 0, 1.0), (-1, 0, 0.5)],                                               # infeasible pair
        [(1, 0, 1.0), (-1, 0, -1.0)],                                              # equality x = 1
        [(0, 0, 1.0)],                                                             # 0.u >= 1
        [(0, 0, -1.0), (1, 1, 1.0)],                                               # empty row
        [(1, 0, 1.0), (0, 1, 2.0), (-1, -1, -3.5)],                                # triangle, vertex active
        [(1, 0, 1.0), (0, 1, 2.0), (-1, -1, -2.9)],                                # empty triangle
        [(1, 1e-3, 1.0), (1, -1e-3, 1.0)],                                         # nearly parallel
        [(1e-6, 0, 1e-6), (0, 1e6, -1e6)],                                         # badly scaled
        [(np.cos(t), np.sin(t), 1.0) for t in np.linspace(0, np.pi / 2, 16)],     # 16 tangents to a circle
    ]
    M = max(len(c) for c in cases)
    rows = np.zeros((M, len(cases), 4))
    rows[:, :, 2] = -1.0
    for a, c in enumerate(cases):
        for m, (a0, a1, b) in enumerate(c):
            rows[m, a, :3] = (a0, a1, b)
    return rows


@pytest.mark.parametrize("lanes", ALL_LANES)
def test_qp_stage_adversarial_rows(oracle, lanes):
    rows = _adversarial_rows()
    A = rows.shape[1]
    un = np.zeros((A, 2))
    un[5] = [0.2, 0.1]
    p = oracle.make_params(R=(1, 0, 0, 1))
    uo, so = oracle.qp_batch(p, rows, un)
    f = _filter(lanes=lanes, R=np.eye(2))
    r = torch.from_numpy(rows).to(_dev())
    ug, sg = f.solve_qp(r, torch.from_numpy(un).to(_dev()))
    torch.cuda.synchronize()
    ug, sg = ug.cpu().numpy(), sg.cpu().numpy()
    assert (sg == so).all(), (sg, so)
    assert np.abs(ug - uo).max() <= 1e-6 * (1 + np.abs(uo).max())


@pytest.mark.parametrize("dtype,b2", [(torch.float64, 1e-10), (torch.float32, 5e-4)], ids=["f64", "f32"])
@pytest.mark.parametrize("lanes", ALL_LANES)
def test_impossible_row_behind_a_far_optimum(oracle, lanes, dtype, b2):
    """0 . u >= b2 > 0 admits no u, however small b2 is against the size of the optimum of the other
    rows (here 479 away, where a unit-normal row would be granted that much rounding slack)."""
    rows = np.zeros((2, 3, 4))
    rows[0, :, :3] = (-0.0452329, 0.0, 21.6777)
    rows[1, 0, :3] = (0.0, 0.0, b2)          # impossible
    rows[1, 1, :3] = (0.0, 0.0, -b2)         # always true
    rows[1, 2, :3] = (0.0, 0.0, 0.0)         # always true
    un = np.tile([-1.7, 0.4], (3, 1))
    _, so = oracle.qp_batch(oracle.make_params(R=(0.5, 0, 0, 0.5)), rows, un)
    f = _filter(lanes=lanes)
    ug, sg = f.solve_qp(torch.from_numpy(rows).to(_dev()).to(dtype), torch.from_numpy(un).to(_dev()).to(dtype))
    torch.cuda.synchronize()
    assert so.tolist() == [1, 0, 0] and sg.cpu().tolist() == [1, 0, 0]
    assert abs(float(ug[1, 0]) + 21.6777 / 0.0452329) < 1e-2


@pytest.mark.parametrize("lanes", ALL_LANES)
@pytest.mark.parametrize("M", [0, 1, 2, 3, 5, 8, 9, 16, 17, 31, 32, 33, 64])
def test_row_counts_and_ragged_batches(oracle, lanes, M):
    """Every row count around the register-tile boundaries; batch sizes that do not fill
    a wavefront or a workgroup; per-agent kinds with empty slots."""
    if lanes not in (0, 32, 64) and M > 16 * lanes:
        pytest.skip("this variant holds at most 16 rows per lane")
    for A in (1, 3, 63, 65, 257):
        rng = np.random.default_rng(1000 + 7 * M + A)
        w = scenarios.monte_carlo(A, (M + 1) // 2, M // 2, int(rng.integers(1 << 30)), nominal="uniform") \
            if M > 0 else dict(state=rng.normal(0, 1, (A, 4)), u_ref=rng.normal(0, 1, (A, 2)),
                               obs=np.zeros((0, 2, A, 4)), obs_type=np.zeros(0, dtype=np.uint8))
        kinds = np.repeat(w["obs_type"][:, None], A, axis=1)
        if M > 0:
            kinds[rng.uniform(size=kinds.shape) < 0.3] = 255       # ragged: empty slots per agent
        w["obs_type"] = np.ascontiguousarray(kinds)
        uo, so, ro = _orc_solve(oracle, w, extra_flags=oracle.FLAG_TYPE_PER_AGENT)
        ug, sg, rg = _gpu_solve(w, torch.float64, lanes)
        assert (sg == so).all()
        assert np.abs(ug - uo).max() <= 1e-8 * (1 + np.abs(uo).max())
        if M == 0:
            assert np.array_equal(ug, w["u_ref"])


def test_shared_obstacle_table(oracle):
    w = scenarios.config4(A=1000, nominal="uniform")
    w["obs"] = np.ascontiguousarray(w["obs"][:, :, 17, :])        # (M,2,4): one table for all
    w["state"][:, :2] = w["state"][17, :2] + np.random.default_rng(0).normal(0, 3, (1000, 2))
    p = oracle.make_params(R=(0.5, 0, 0, 0.5), flags=oracle.FLAG_OBS_SHARED | oracle.FLAG_SLIP_IO)
    uo, so, _ = oracle.solve_batch(p, w["state"], w["u_ref"], w["obs"], w["obs_type"])
    for lanes in (0, 1, 4, 64):
        ug, sg = _gpu_solve(w, torch.float64, lanes, want_rows=False)
        assert (sg == so).all() and np.abs(ug - uo).max() <= 1e-8 * (1 + np.abs(uo).max())


@pytest.mark.parametrize("lanes", ALL_LANES)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_tiled_obstacle_layout(oracle, lanes, dtype):
    """obs as tiles of 64 agents, (ceil(A/64), M, 2, 64, 4): the bits of the planes layout for
    every kernel variant, batch sizes with a partial last tile included, and the oracle's answer."""
    from sccav_cbf_amd.batched import pack_obs, pack_obs_tiled, planes_to_tiled
    for A, M in ((1, 3), (63, 16), (64, 8), (65, 16), (1000, 16), (4097, 5), (130, 33)):
        if lanes not in (0, 32, 64) and M > 16 * lanes:
            continue
        w = scenarios.monte_carlo(A, (M + 1) // 2, M // 2, 4242 + A, nominal="uniform")
        t = _to_dev(w, dtype)
        tiles = planes_to_tiled(t["obs"])
        assert tiles.shape == ((A + 63) // 64, M, 2, 64, 4)
        slots = t["obs"].permute(0, 2, 1, 3).reshape(M, A, 8)
        assert torch.equal(pack_obs(slots), t["obs"]) and torch.equal(pack_obs_tiled(slots), tiles)
        f = _filter(lanes=lanes, slip_io=True)
        up, sp, rp = f.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], want_rows=True)
        ut, st, rt = f.solve(t["state"], t["u_ref"], tiles, t["obs_type"], want_rows=True)
        assert torch.equal(up, ut) and torch.equal(sp, st) and torch.equal(rp, rt)
        if dtype == torch.float64:
            uo, so, ro = _orc_solve(oracle, w)
            assert (st.cpu().numpy() == so).all()
            assert np.abs(ut.cpu().numpy() - uo).max() <= 1e-8 * (1 + np.abs(uo).max())
    # the partials entry point reads the same layout
    lib = _lib.load()
    w = scenarios.monte_carlo(130, 3, 2, 5, nominal="uniform")
    t = _to_dev(w, dtype)
    outs = []
    for obs, fl in ((t["obs"], 0), (planes_to_tiled(t["obs"]), _lib.FLAG_OBS_TILED)):
        out = torch.empty((5, 2, 130, 4), dtype=dtype, device=_dev())
        p = _lib.Params()
        lib.sccbf_default_params(C.byref(p))
        p.flags = fl
        fn = lib.sccbf_partials_f64 if dtype == torch.float64 else lib.sccbf_partials_f32
        _lib.check(fn(C.byref(p), 130, 5, t["state"].data_ptr(), obs.data_ptr(), t["obs_type"].data_ptr(),
                      out.data_ptr(), None))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    # one table for all agents has no tiled form
    p.flags = _lib.FLAG_OBS_TILED | _lib.FLAG_OBS_SHARED
    assert fn(C.byref(p), 130, 5, t["state"].data_ptr(), obs.data_ptr(), t["obs_type"].data_ptr(),
              out.data_ptr(), None) == _lib.ERR_ARG


def test_nonfinite_inputs_are_flagged(oracle):
    w = scenarios.config4(A=300, nominal="uniform")
    w["state"][5, 0] = np.nan
    w["u_ref"][9, 1] = np.inf
    w["obs"][3, 0, 20, 2] = np.nan
    w["obs"][12, 0, 31, 0] = -np.inf        # cone centre: the row becomes non-finite
    w["obs"][2, 1, 40, 3] = np.nan           # unused scalar of an ellipse slot: ignored
    uo, so, _ = _orc_solve(oracle, w)
    assert list(np.where(so == 2)[0]) == [5, 9, 20, 31]
    for dtype in (torch.float64, torch.float32):
        for lanes in (1, 4, 64):
            ug, sg = _gpu_solve(w, dtype, lanes, want_rows=False)
            assert (sg == so).all()
            assert np.allclose(ug[so == 2], w["u_ref"][so == 2], equal_nan=True)


def _many_rows(A, M, seed):
    """The 16-slot scene of configuration 4 repeated up to M slots: copy k of an obstacle stands 10 k ... 10 k + 10 m
    farther from the agent along the line of sight, so the feasible share stays that of 16 rows."""
    base = scenarios.config4(A=A, nominal="uniform", seed=seed)
    reps = -(-M // 16)
    rng = np.random.default_rng(seed)
    obs = np.concatenate([base["obs"]] * reps, axis=0)[:M].copy()
    los = obs[:, 0, :, 0:2] - base["state"][None, :, 0:2]
    los /= np.maximum(np.linalg.norm(los, axis=2, keepdims=True), 1e-3)
    far = (np.arange(M) // 16)[:, None, None] * 10.0
    obs[:, 0, :, 0:2] += los * (far + 10.0 * rng.uniform(size=(M, 1, 1)) * (far > 0))
    return dict(state=base["state"], u_ref=base["u_ref"], obs=obs, obs_type=np.tile(base["obs_type"], reps)[:M].copy())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("M,lanes", [(_lib.MAX_ROWS, 0), (800, 32), (_lib.MAX_ROWS, 64), (400, 16), (257, 16),
                                     (300, 0), (700, 0), (129, 0), (900, 0), (150, 0), (200, 16), (129, 32)])
def test_rows_in_lds_only_up_to_the_maximum_row_count(oracle, M, lanes, dtype):
    """More rows than the register-resident kernels take, up to M = SCCBF_MAX_ROWS: rows staged in LDS only,
    16 or 32 lanes or a whole wavefront per QP; batches that end inside a workgroup and inside a pair of them."""
    for A in (9, 333):
        w = _many_rows(A, M, 77 + A)
        uo, so, ro = _orc_solve(oracle, w)
        ug, sg = _gpu_solve(w, dtype, lanes, want_rows=False)
        _check(ug, sg, uo, so, ro, w["u_ref"], dtype)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("M,lanes", [(9, 1), (12, 1), (20, 2), (40, 4), (48, 4), (33, 4), (40, 8), (48, 8), (72, 8),
                                     (96, 8), (100, 8), (65, 16), (80, 16), (96, 16), (112, 16), (48, 0), (96, 0)])
def test_row_counts_that_leave_register_rows_unused(oracle, M, lanes, dtype):
    """Rows in registers, 8 or 16 per lane, of which the last ones lie past M for every lane of a group: those
    are empty rows whose arithmetic and scan are skipped.  Answers and rows against the oracle, and the QP stage
    alone on the same rows."""
    A = 333
    w = _many_rows(A, M, 500 + M)
    uo, so, ro = _orc_solve(oracle, w)
    ug, sg, rg = _gpu_solve(w, dtype, lanes)
    _check(ug, sg, uo, so, ro, w["u_ref"], dtype)
    tol = 1e-9 if dtype == torch.float64 else 1e-3
    assert (np.abs(rg - ro) / (np.abs(ro).max(axis=2, keepdims=True) + 1.0)).max() <= tol
    if dtype == torch.float64:
        dev = _dev()
        f = _filter(lanes=lanes, slip_io=True)
        rows = np.ascontiguousarray(ro)
        uq, sq = f.solve_qp(torch.from_numpy(rows).to(dev), torch.from_numpy(w["u_ref"]).to(dev))
        torch.cuda.synchronize()
        uqo, sqo = oracle.qp_batch(oracle.make_params(R=R_HALF.reshape(-1)), rows, w["u_ref"])
        assert (sq.cpu().numpy() == sqo).all()
        assert np.abs(uq.cpu().numpy() - uqo).max() <= 1e-8 * (1 + np.abs(uqo).max())


def test_beyond_the_maximum_row_count_is_refused():
    M = _lib.MAX_ROWS
    w = scenarios.monte_carlo(3, M // 2, M // 2, 77, nominal="uniform")
    with pytest.raises(_lib.SccbfError):     # sixteen fp64 tables of 1024 rows are more LDS than a CU has
        _gpu_solve(w, torch.float64, 16, want_rows=False)
    f = _filter()
    t = _to_dev(scenarios.monte_carlo(2, M // 2 + 1, M // 2, 1, nominal="uniform"), torch.float64)
    with pytest.raises(_lib.SccbfError):
        f.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"])


def test_full_size_batch_properties(oracle):
    """C4 at full size (65536 x 16, fp32): shard invariance, feasibility of every answer
    against the kernel's own rows, idempotence of the projection, and the oracle on a
    4096-agent sample."""
    w = scenarios.config4(A=65536, nominal="uniform")
    t = _to_dev(w, torch.float32)
    f = _filter(slip_io=True)
    u, st, rows = f.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], want_rows=True)
    # (i) two shards give the bits of the full batch
    h = 65536 // 2
    for lo in (0, h):
        us, ss = f.solve(t["state"][lo:lo + h].contiguous(), t["u_ref"][lo:lo + h].contiguous(),
                         t["obs"][:, :, lo:lo + h, :].contiguous(), t["obs_type"])
        assert torch.equal(us, u[lo:lo + h]) and torch.equal(ss, st[lo:lo + h])
    # (ii) feasibility in fp64 on the device
    r64, u64 = rows.double(), u.double()
    lhs = r64[:, :, 0] * u64[None, :, 0] + r64[:, :, 1] * u64[None, :, 1]
    scale = torch.hypot(r64[:, :, 0], r64[:, :, 1]) * torch.hypot(u64[:, 0], u64[:, 1])[None, :] + r64[:, :, 2].abs()
    viol = ((r64[:, :, 2] - lhs) / (scale + 1e-30)).max(dim=0).values
    ok = st == 0
    assert ok.sum().item() > 30000
    assert viol[ok].max().item() <= 1e-4
    # (iii) a feasible answer is its own projection
    #       (answers ~1e4 away from u_ref sit in a wedge of two rows that are parallel to
    #       within 1e-4 rad; fp32 cannot even represent a feasible point there: excluded)
    u2, st2 = f.solve(t["state"], u.contiguous(), t["obs"], t["obs_type"])
    sane = ok & ((u - t["u_ref"]).abs().max(dim=1).values < 1000.0)
    assert sane.sum().item() >= 0.98 * ok.sum().item()
    assert (st2[sane] == 0).all()
    assert ((u2 - u)[sane].abs().max(dim=1).values / (1 + u[sane].abs().max(dim=1).values)).max().item() <= 2e-4
    # (iv) oracle on a sample
    idx = np.random.default_rng(3).choice(65536, 4096, replace=False)
    ws = dict(state=w["state"][idx], u_ref=w["u_ref"][idx], obs=np.ascontiguousarray(w["obs"][:, :, idx, :]),
              obs_type=w["obs_type"])
    uo, so, ro = _orc_solve(oracle, ws)
    _check(u.double().cpu().numpy()[idx], st.cpu().numpy()[idx], uo, so, ro, ws["u_ref"], torch.float32)


@pytest.mark.parametrize("alpha", [0.5, 1.0, 5.0])
def test_fp32_vs_fp64_tolerance_sweep(oracle, alpha):
    """C5: lanes + cones, 16384 agents: fp32 against fp64 on the GPU, and fp64 against the oracle."""
    w = scenarios.lanes_and_cones(A=16384)
    u64, s64, r64 = _gpu_solve(w, torch.float64, 0, alpha=alpha)
    u32, s32, r32 = _gpu_solve(w, torch.float32, 0, alpha=alpha)
    uo, so, ro = _orc_solve(oracle, w, alpha=alpha)
    _check(u64, s64, uo, so, ro, w["u_ref"], torch.float64)
    _check(u32, s32, u64, s64, r64, w["u_ref"], torch.float32)
    both = (s64 == 0) & (s32 == 0)
    du = np.abs(u32 - u64).max(axis=1)[both]
    print("alpha %.1f: max|du| %.3e  p99.9 %.3e  status agreement %.5f" % (
        alpha, du.max(), np.quantile(du, 0.999), (s32 == s64).mean()))


@pytest.mark.parametrize("layout", ["planes", "tiles"])
def test_batch_beyond_4_gib_of_obstacles(oracle, layout):
    """9 M agents x 16 slots (fp32): the obstacle array is 4.6 GB, so slot offsets exceed 32 bits.
    Inputs are made on the device; the oracle checks agents sampled over the whole range,
    the last ones included."""
    from sccav_cbf_amd.batched import planes_to_tiled
    A, M = 9_000_000, 16
    g = torch.Generator(device=_dev()).manual_seed(7)

    def U(shape, lo, hi):
        return torch.rand(shape, generator=g, device=_dev(), dtype=torch.float32) * (hi - lo) + lo

    state = torch.stack([U((A,), -50, 50), U((A,), -50, 50), U((A,), -3.1, 3.1), U((A,), 0, 15)], dim=1).contiguous()
    u_ref = torch.stack([U((A,), -2, 2), U((A,), -0.4, 0.4)], dim=1).contiguous()
    obs = torch.zeros((M, 2, A, 4), dtype=torch.float32, device=_dev())
    for m in range(M):
        rng_, brg = U((A,), 15, 60), U((A,), -3.1, 3.1)
        obs[m, 0, :, 0] = state[:, 0] + rng_ * torch.cos(brg)
        obs[m, 0, :, 1] = state[:, 1] + rng_ * torch.sin(brg)
        if m < 8:      # ellipse: cx, cy, a, b | theta, vx, vy
            obs[m, 0, :, 2], obs[m, 0, :, 3] = U((A,), 2, 20), U((A,), 1, 10)
            obs[m, 1, :, 0], obs[m, 1, :, 1], obs[m, 1, :, 2] = U((A,), -3.1, 3.1), U((A,), -2, 2), U((A,), -2, 2)
        else:          # cone: cx, cy, theta_o, v_o | a, beta
            obs[m, 0, :, 2], obs[m, 0, :, 3] = U((A,), -3.1, 3.1), U((A,), 0, 5)
            obs[m, 1, :, 0] = U((A,), 2.5, 6.5)
    ty = torch.tensor([0] * 8 + [1] * 8, dtype=torch.uint8, device=_dev())
    idx = torch.cat([torch.randint(0, A, (2000,), generator=g, device=_dev()),
                     torch.arange(A - 70, A, device=_dev()), torch.arange(0, 70, device=_dev())])
    w = dict(state=state[idx].double().cpu().numpy(), u_ref=u_ref[idx].double().cpu().numpy(),
             obs=np.ascontiguousarray(obs[:, :, idx, :].double().cpu().numpy()), obs_type=ty.cpu().numpy())
    if layout == "tiles":
        obs = planes_to_tiled(obs)
    f = _filter(slip_io=True)
    u, st = f.solve(state, u_ref, obs, ty)
    torch.cuda.synchronize()
    del obs
    uo, so, ro = _orc_solve(oracle, w)
    _check(u[idx].double().cpu().numpy(), st[idx].cpu().numpy(), uo, so, ro, w["u_ref"], torch.float32)


def _mixed_kinds(A=1500, seed=77):
    """Every kind side by side in one batch, kinds per agent: lanes of a wavefront hold different kinds."""
    w = scenarios.config4(A=A, nominal="uniform", seed=seed)
    wp = scenarios.prepared(w)
    rng = np.random.default_rng(seed)
    M = w["obs"].shape[0]
    pick = rng.random((M, A)) < 0.5
    obs = np.where(pick[:, None, :, None], wp["obs"], w["obs"])
    ty = np.where(pick, wp["obs_type"][:, None], w["obs_type"][:, None]).astype(np.uint8)
    ty[rng.random((M, A)) < 0.1] = 255
    return dict(state=w["state"], u_ref=w["u_ref"], obs=np.ascontiguousarray(obs), obs_type=np.ascontiguousarray(ty))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("lanes", (0, 1, 4, 16, 64))
def test_prepared_and_written_slots_side_by_side(oracle, dtype, lanes):
    w = _mixed_kinds()
    uo, so, ro = _orc_solve(oracle, w, extra_flags=oracle.FLAG_TYPE_PER_AGENT)
    ug, sg, rg = _gpu_solve(w, dtype, lanes)
    _check(ug, sg, uo, so, ro, w["u_ref"], dtype)
    tol = 1e-9 if dtype == torch.float64 else 1e-3
    assert (np.abs(rg - ro) / (np.abs(ro).max(axis=2, keepdims=True) + 1.0)).max() <= tol


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("layout", ["planes", "tiles", "shared", "per_agent_types"])
def test_prepare_obs_on_the_device_equals_the_host_restatement(dtype, layout):
    """sccbf_prepare_obs_* rewrites slots and kinds in place; checked against scenarios.prepared."""
    from sccav_cbf_amd.batched import planes_to_tiled, prepare_obs
    A = 1 if layout == "shared" else 333
    w = scenarios.lanes_and_cones(A=A, n_cone=5)
    e = scenarios.config3(A=A)
    w["obs"] = np.concatenate([w["obs"], e["obs"][:3]], axis=0)          # lanes, cones, ellipses
    w["obs_type"] = np.concatenate([w["obs_type"], e["obs_type"][:3]])
    if layout == "per_agent_types":
        ty = np.repeat(w["obs_type"][:, None], A, axis=1)
        ty[:, ::7] = 255
        w["obs_type"] = np.ascontiguousarray(ty)
    t = _to_dev(w, dtype)
    host = {k: (v.cpu().double().numpy() if v.dtype.is_floating_point else v.cpu().numpy()) for k, v in t.items()}
    want = scenarios.prepared(host)                     # from the values the device holds
    obs, ty = t["obs"], t["obs_type"]
    if layout == "tiles":
        obs = planes_to_tiled(obs)
    elif layout == "shared":
        obs = obs[:, :, 0, :].contiguous()
    prepare_obs(obs, ty, agents=A)
    torch.cuda.synchronize()
    got = obs
    if layout == "tiles":
        M = obs.shape[1]
        got = obs.permute(1, 2, 0, 3, 4).reshape(M, 2, -1, 4)[:, :, :A]
    elif layout == "shared":
        got = obs[:, :, None, :]
    got = got.double().cpu().numpy()
    assert np.array_equal(ty.cpu().numpy(), want["obs_type"])
    rel = np.abs(got - want["obs"]) / (np.abs(want["obs"]) + 1e-30)
    assert rel.max() <= (1e-15 if dtype == torch.float64 else 6e-8), rel.max()   # one rounding to T
    # a second call finds nothing left to do
    before = obs.clone()
    prepare_obs(obs, ty, agents=A)
    torch.cuda.synchronize()
    assert torch.equal(before, obs)
