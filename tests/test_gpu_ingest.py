"""Bounding boxes -> slots and lane points -> lane slots on the GPU against the oracle, and the
whole chain perception -> slots -> filter without leaving the device."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sccav_cbf_amd import _lib  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled  # noqa: E402
from sccav_cbf_amd.ingest import ingest_bbox, lane_fit  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _boxes(A, M, seed):
    rng = np.random.default_rng(seed)
    bbox = np.zeros((M, 2, A, 4))
    bbox[:, 0, :, 0] = rng.uniform(-60, 60, (M, A))
    bbox[:, 0, :, 1] = rng.uniform(-60, 60, (M, A))
    bbox[:, 0, :, 2] = rng.uniform(1.5, 3.0, (M, A))
    bbox[:, 0, :, 3] = rng.uniform(0.7, 1.2, (M, A))
    bbox[:, 1, :, 0] = rng.uniform(-np.pi, np.pi, (M, A))
    bbox[:, 1, :, 1] = rng.uniform(0, 8, (M, A))
    return bbox, rng


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("per_agent", [False, True])
@pytest.mark.parametrize("A", [1, 63, 1000])
def test_ingest_bbox_matches_oracle(oracle, dtype, per_agent, A):
    M = 7
    bbox, rng = _boxes(A, M, 3)
    kind = rng.choice(np.array([0, 1, 4, 5, 255], dtype=np.uint8), (M, A) if per_agent else (M,))   # 4, 5: prepared forms
    exp = oracle.ingest_bbox(bbox, kind, buffer=0.7)
    b = torch.from_numpy(bbox).to(DEV).to(dtype)
    k = torch.from_numpy(kind).to(DEV)
    got = ingest_bbox(b, k, buffer=0.7)
    tiled = ingest_bbox(b, k, buffer=0.7, tiled=True)
    torch.cuda.synchronize()
    tol = 1e-14 if dtype == torch.float64 else 1e-6
    assert np.abs(got.double().cpu().numpy() - exp).max() <= tol * 100
    # the tiled output holds the same slots (pad lanes of the last tile stay zero)
    assert torch.equal(planes_to_tiled(got), tiled)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("degree,K", [(1, 24), (2, 24), (3, 24), (0, 7), (3, 8), (2, 33), (3, 64), (1, 65), (3, 150)])
def test_lane_fit_matches_oracle(oracle, dtype, degree, K):
    """The point counts cover every form of the kernel: one lane per agent holding all points (K <= 8),
    four lanes holding 8 or 16 each (K <= 32, 64), and lists read again on every pass (K > 64)."""
    A, M = 500, 3
    rng = np.random.default_rng(10 + degree + 100 * (K != 24) * K)
    x0 = rng.uniform(-80, 80, A)
    px = x0[None, :] + np.sort(rng.uniform(-5, 60, (K, A)), axis=0)
    true = rng.normal(0, [5, 0.3, 0.01, 2e-4], (A, 4))[:, :degree + 1]
    py = sum(true[None, :, j] * (px - x0[None, :]) ** j for j in range(degree + 1)) + rng.normal(0, 0.1, (K, A))
    sigma = np.full((K, A), 10.0)
    sigma[0, :] = 0.01                                      # a fixed point at the near end
    sigma[K - 3:, A // 2:] = 0.0                            # ragged: half of the agents have 3 points fewer
    if dtype == torch.float32:                              # what the device sees
        px, py, sigma = (v.astype(np.float32).astype(np.float64) for v in (px, py, sigma))
    exp = oracle.lane_fit(px, py, sigma, degree=degree)
    obs = torch.full((M, 2, A, 4), 7.0, dtype=dtype, device=DEV)
    t = lambda v: torch.from_numpy(v).to(DEV).to(dtype).contiguous()       # noqa: E731
    lane_fit(t(px), t(py), obs, 1, sigma=t(sigma), degree=degree, buffer=1.25)
    torch.cuda.synchronize()
    o = obs.double().cpu().numpy()
    assert (o[0] == 7.0).all() and (o[2] == 7.0).all()      # only the named slot is written
    assert np.allclose(o[1, 1], np.array([1.25, 0, 0, 0])[None, :])
    c = o[1, 0]                                             # (A, 4)
    # same curve over the span of the points (raw-power coefficients of far-away x are ill-conditioned)
    xs = np.linspace(px.min(axis=0), px.max(axis=0), 20)
    val = lambda cc: sum(cc[None, :, j] * xs ** j for j in range(4))       # noqa: E731
    # fp32: the coefficient of x^3 at |x| ~ 100 in single precision
    tol = 1e-9 if dtype == torch.float64 else 2e-2
    assert np.abs(val(c) - val(exp)).max() <= tol
    if dtype == torch.float64:
        assert np.abs(c - exp).max() <= 1e-8 * (1 + np.abs(exp).max())
    assert (c[:, degree + 1:] == 0).all()


def test_lane_fit_known_answer_and_underdetermined():
    """test_scripts/lane_cbf_test.py:199-200 (the cubic through four points) for every agent; an agent
    with three usable points cannot fix a cubic: NaN, which the filter reports as status 2."""
    x = np.array([-1.371, -0.75, 0, 0.333])
    y = np.array([0, 6.938, 3, 1.852])
    A = 5
    px = torch.from_numpy(np.repeat(x[:, None], A, axis=1)).to(DEV).contiguous()
    py = torch.from_numpy(np.repeat(y[:, None], A, axis=1)).to(DEV).contiguous()
    sigma = torch.full((4, A), 10.0, dtype=torch.float64, device=DEV)
    sigma[3, 4] = 0.0
    obs = torch.zeros((1, 2, A, 4), dtype=torch.float64, device=DEV)
    lane_fit(px, py, obs, 0, sigma=sigma)
    torch.cuda.synchronize()
    c = obs[0, 0].cpu().numpy()
    exp = np.linalg.solve(np.vander(x, 4, increasing=True), y)
    assert np.allclose(c[:4], exp[None, :], rtol=1e-10)
    assert np.isnan(c[4]).all()
    f = BatchedCBF()
    state = torch.zeros((A, 4), dtype=torch.float64, device=DEV)
    u, st = f.solve(state, torch.zeros((A, 2), dtype=torch.float64, device=DEV), obs,
                    torch.tensor([_lib.OBS_LANE], dtype=torch.uint8, device=DEV))
    assert st.cpu().tolist() == [0, 0, 0, 0, 2]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_perception_to_filter_on_the_device(oracle, dtype):
    """Boxes and lane points in, safe inputs out: ingest_bbox + two lane_fit calls + solve on the GPU
    against the oracle doing the same on the host (the per-tick work of the CARLA drivers,
    multi_obstacle_CBF_local_with_lanes.py, for a batch)."""
    A, K = 2048, 16
    bbox, rng = _boxes(A, 6, 21)
    state = np.stack([rng.uniform(-5, 5, A), rng.uniform(-1, 1, A), rng.normal(0, 0.1, A), rng.uniform(3, 12, A)], axis=1)
    bbox[:, 0, :, 0] = state[None, :, 0] + rng.uniform(8, 50, (6, A))
    bbox[:, 0, :, 1] = state[None, :, 1] + rng.uniform(-6, 6, (6, A))
    kind6 = np.array([0, 0, 1, 1, 1, 255], dtype=np.uint8)
    px = state[None, :, 0] + np.linspace(-5, 60, K)[:, None] + rng.normal(0, 0.3, (K, A))
    left = 4.0 + 0.01 * (px - state[None, :, 0]) + rng.normal(0, 0.05, (K, A))
    right = -4.0 + 0.01 * (px - state[None, :, 0]) + rng.normal(0, 0.05, (K, A))
    u_ref = np.stack([rng.normal(0, 0.5, A), rng.normal(0, 0.05, A)], axis=1)
    if dtype == torch.float32:
        bbox, state, px, left, right, u_ref = (v.astype(np.float32).astype(np.float64)
                                               for v in (bbox, state, px, left, right, u_ref))
    # host
    obs = np.zeros((8, 2, A, 4))
    obs[:6] = oracle.ingest_bbox(bbox, kind6, buffer=0.5)
    for m, pts in ((6, left), (7, right)):
        obs[m, 0] = oracle.lane_fit(px, pts, degree=2)
        obs[m, 1, :, 0] = 1.5
    kind = np.concatenate([kind6, [_lib.OBS_LANE, _lib.OBS_LANE]]).astype(np.uint8)
    p = oracle.make_params(alpha=1.0, R=(0.5, 0, 0, 0.5))
    uo, so, ro = oracle.solve_batch(p, state, u_ref, obs, kind, want_rows=True)
    # device
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(DEV).to(dtype)     # noqa: E731
    kd = torch.from_numpy(kind).to(DEV)
    bb8 = torch.zeros((8, 2, A, 4), dtype=dtype, device=DEV)
    bb8[:6] = t(bbox)
    kind_in = torch.from_numpy(np.concatenate([kind6, [255, 255]]).astype(np.uint8)).to(DEV)
    for tiled in (False, True):
        og = ingest_bbox(bb8, kind_in, buffer=0.5, tiled=tiled)
        lane_fit(t(px), t(left), og, 6, degree=2, buffer=1.5)
        lane_fit(t(px), t(right), og, 7, degree=2, buffer=1.5)
        ug, sg = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5])).solve(t(state), t(u_ref), og, kd)
        torch.cuda.synchronize()
        ug, sg = ug.double().cpu().numpy(), sg.cpu().numpy()
        assert (so == 0).mean() > 0.5 and (np.abs(uo - u_ref).max(axis=1) > 1e-3).sum() > 100
        if dtype == torch.float64:
            assert np.array_equal(sg, so)
            assert (np.abs(ug - uo).max(axis=1) / np.maximum(1, np.abs(uo).max(axis=1))).max() <= 1e-7
        else:
            assert (sg != so).mean() <= 2e-3
            ok = (sg == 0) & (so == 0) & (np.abs(uo[:, 1]) < 1.2)
            err = np.abs(ug - uo).max(axis=1) / np.maximum(1, np.abs(uo).max(axis=1))
            assert np.quantile(err[ok], 0.99) <= 2e-3


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", ["dbm", "kbm", "dum", "sadbm"])
@pytest.mark.parametrize("M", [6, 40, 150])
def test_boxes_as_slots_equal_ingestion_then_filter(oracle, dtype, dyn, M):
    """Kinds 6, 7 (the slot is the bounding box + buffer, no ingestion pass): the answers of ingest_bbox followed by
    the filter, in both layouts and for every model and kernel family: ellipse rows bit for bit, cone rows to the last
    few bits (the compiler fuses multiply-adds differently where it knows the cone's heading to be zero), the inputs
    accordingly; and the oracle's answers."""
    from sccav_cbf_amd.ingest import bbox_as_obs
    from tests import parity
    A = 1500
    bbox, rng = _boxes(A, M, 33 + M)
    state = np.stack([rng.uniform(-5, 5, A), rng.uniform(-1, 1, A), rng.normal(0, 0.3, A), rng.uniform(3, 12, A)], axis=1)
    bbox[:, 0, :, 0] = state[None, :, 0] + rng.uniform(8, 60, (M, A))
    bbox[:, 0, :, 1] = state[None, :, 1] + rng.uniform(-8, 8, (M, A))
    kind = rng.choice(np.array([0, 1, 255], dtype=np.uint8), M, p=[0.45, 0.45, 0.1])
    u_ref = np.stack([rng.normal(0, 0.5, A) + (6.0 if dyn == "kbm" else 0.0), rng.normal(0, 0.05, A)], axis=1)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(DEV).to(dtype)     # noqa: E731
    kd = torch.from_numpy(kind).to(DEV)
    f = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), dynamics=dyn)
    kw = lambda: dict(slip=torch.full((A, 2), 0.05, dtype=dtype, device=DEV), dt=0.05) if dyn == "sadbm" else {}   # noqa: E731
    boxes, kb = bbox_as_obs(t(bbox), kd, buffer=0.6)
    assert set(kb.cpu().tolist()) <= {6, 7, 255}
    eps = 1e-13 if dtype == torch.float64 else 1e-6
    for tiled in (False, True):
        slots = ingest_bbox(t(bbox), kd, buffer=0.6, tiled=tiled)
        k0, k1 = kw(), kw()
        u0, s0, r0 = f.solve(t(state), t(u_ref), slots, kd, want_rows=True, **k0)
        u1, s1, r1 = f.solve(t(state), t(u_ref), planes_to_tiled(boxes) if tiled else boxes, kb, want_rows=True, **k1)
        torch.cuda.synchronize()
        assert torch.equal(r0[kd == 0], r1[kd == 0])
        scale = r0.abs().amax(dim=2, keepdim=True) + 1.0
        # (a cone that closes at a small relative speed divides by it: a last-bit difference upstream is magnified)
        dr = ((r0 - r1).abs() / scale).double().cpu().numpy()
        assert np.quantile(dr, 0.999) <= 32 * eps and dr.max() <= (1e-10 if dtype == torch.float64 else 1e-3)
        assert int((s0 != s1).sum()) <= A // 1000
        r0n, u0n, u1n = r0.double().cpu().numpy(), u0.double().cpu().numpy(), u1.double().cpu().numpy()
        # a vertex of two rows moves by (row difference) / |sin| of their angle: rows a few ulp apart, |sin| >= 1e-2
        well = ((s0 == 0) & (s1 == 0)).cpu().numpy() & (parity.active_det(u0n, r0n) >= 1e-2)
        du = np.abs(u0n - u1n).max(axis=1) / np.maximum(1.0, np.abs(u0n).max(axis=1))
        assert well.sum() > 50
        if dtype == torch.float64:
            assert du[well].max() <= 1e-9
        else:   # (the conditioning estimate is in the QP's own coordinates only for slip-angle inputs: a quantile)
            assert np.quantile(du[well], 0.99) <= 1e-4 and du[well].max() <= 0.1
        if dyn == "sadbm":
            assert float((k0["slip"] - k1["slip"]).abs()[torch.from_numpy(well).to(DEV)].max()) <= 1e3 * eps
    if dyn in ("dbm", "kbm") and dtype == torch.float64:
        boxed = bbox.copy()
        boxed[:, 1, :, 2] = 0.6
        p = oracle.make_params(alpha=1.0, R=(0.5, 0, 0, 0.5), dynamics=oracle.DYN_KBM if dyn == "kbm" else oracle.DYN_DBM)
        uo, so, _ = oracle.solve_batch(p, state, u_ref, boxed, kb.cpu().numpy(), want_rows=True)
        assert np.array_equal(s1.cpu().numpy(), so)
        assert (np.abs(u1.cpu().numpy() - uo).max(axis=1) / np.maximum(1, np.abs(uo).max(axis=1))).max() <= 1e-7
