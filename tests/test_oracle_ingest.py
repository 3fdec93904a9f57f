"""The oracle's restatements of lane fitting and bounding-box ingestion against the reference's own
fixture and call pattern (scipy.optimize.curve_fit is importable here, cvxpy is not)."""
import numpy as np
import scipy.optimize as sciopt
from numpy.polynomial import Polynomial


def _curve_fit_as_the_reference_calls_it(x, y, n, sigma):
    """The call of cbf/obstacles.py:765-772 (own restatement of those eight lines)."""
    def func(x, *p):
        return Polynomial(p)(x)
    coeffs, _ = sciopt.curve_fit(func, x, y, np.zeros(n + 1), sigma=sigma)
    return coeffs


def test_lane_fit_reference_known_answer(oracle):
    """test_scripts/lane_cbf_test.py:199-200: four points, cubic: the fit interpolates them."""
    x = np.array([-1.371, -0.75, 0, 0.333])
    y = np.array([0, 6.938, 3, 1.852])
    c = oracle.lane_fit(x[:, None], y[:, None])[0]
    assert np.abs(Polynomial(c)(x) - y).max() < 1e-11
    assert np.allclose(c, _curve_fit_as_the_reference_calls_it(x, y, 3, np.full(4, 10.0)), rtol=1e-6, atol=1e-8)
    assert np.allclose(c, np.linalg.solve(np.vander(x, 4, increasing=True), y), rtol=1e-10)


def test_lane_fit_matches_curve_fit(oracle):
    rng = np.random.default_rng(5)
    for case in range(30):
        K, n = int(rng.integers(5, 40)), int(rng.integers(1, 4))
        x = np.sort(rng.uniform(-5, 60, K)) + rng.uniform(-100, 100)
        true = rng.normal(0, [5, 0.3, 0.01, 2e-4])[:n + 1]
        y = Polynomial(true)(x - x.mean()) + rng.normal(0, 0.2, K)
        sigma = np.full(K, 10.0)
        sigma[rng.integers(0, K, 2)] = 0.01          # fixed points: alpha (cbf/obstacles.py:753-756)
        c = oracle.lane_fit(x[:, None], y[:, None], sigma[:, None], degree=n)[0]
        ref = _curve_fit_as_the_reference_calls_it(x, y, n, sigma)
        xs = np.linspace(x.min(), x.max(), 50)
        # curve_fit iterates (Levenberg-Marquardt from zero coefficients, stops at a relative cost
        # change of 1e-8) on raw powers of a far-away x; the oracle solves the same least-squares
        # problem exactly: its cost is never worse, and the two curves agree to the millimetre
        cost = lambda p: (((Polynomial(p)(x) - y) / sigma) ** 2).sum()      # noqa: E731
        assert cost(c[:n + 1]) <= cost(ref) * (1 + 1e-9), case
        assert np.abs(Polynomial(c)(xs) - Polynomial(ref)(xs)).max() <= 5e-3, case
        assert (c[n + 1:] == 0).all()


def test_lane_fit_skips_points_without_weight(oracle):
    x = np.array([0.0, 1.0, 2.0, 3.0, 50.0])[:, None]
    y = np.array([1.0, 3.0, 5.0, 7.0, -1e6])[:, None]
    s = np.array([10.0, 10.0, 10.0, 10.0, 0.0])[:, None]
    assert np.allclose(oracle.lane_fit(x, y, s, degree=1)[0], [1.0, 2.0, 0.0, 0.0], atol=1e-12)


def test_ingest_bbox_by_hand(oracle):
    """One box read as an ellipse and as a cone (cbf/obstacles.py:320-331, 534-543, 146-162, 357)."""
    bbox = np.zeros((3, 2, 1, 4))
    bbox[:, 0, 0] = [10.0, -3.0, 2.0, 1.0]
    bbox[:, 1, 0, 0:2] = [0.4, 3.0]
    obs = oracle.ingest_bbox(bbox, np.array([oracle.OBS_ELLIPSE, oracle.OBS_CONE, oracle.OBS_EMPTY]), buffer=0.5)
    assert np.allclose(obs[0].reshape(-1), [10.0, -3.0, 2.5, 1.5, 0.4, 0, 0, 0])
    assert np.allclose(obs[1].reshape(-1), [10.0, -3.0, 0.0, 3.0, np.hypot(2.0, 1.0) + 0.5, 0, 0, 0])
    assert not obs[2].any()


def test_box_kinds_equal_ingested_slots(oracle):
    """Kinds 6, 7: the slot is the bounding box itself (+ buffer); the rows are those of the obstacle that
    from_bounding_box makes of it (cbf/obstacles.py:320-331, 534-543), for every model."""
    rng = np.random.default_rng(5)
    A, M = 300, 6
    bbox = np.zeros((M, 2, A, 4))
    state = np.stack([rng.uniform(-5, 5, A), rng.uniform(-1, 1, A), rng.normal(0, 0.3, A), rng.uniform(3, 12, A)], axis=1)
    bbox[:, 0, :, 0] = state[None, :, 0] + rng.uniform(6, 40, (M, A))
    bbox[:, 0, :, 1] = state[None, :, 1] + rng.uniform(-6, 6, (M, A))
    bbox[:, 0, :, 2] = rng.uniform(1.5, 3.0, (M, A))
    bbox[:, 0, :, 3] = rng.uniform(0.7, 1.2, (M, A))
    bbox[:, 1, :, 0] = rng.uniform(-np.pi, np.pi, (M, A))
    bbox[:, 1, :, 1] = rng.uniform(0, 8, (M, A))
    kind = np.array([0, 1, 0, 1, 1, 255], dtype=np.uint8)
    slots = oracle.ingest_bbox(bbox, kind, buffer=0.7)
    boxed = bbox.copy()
    boxed[:, 1, :, 2] = 0.7
    kind_box = np.array([6, 7, 6, 7, 7, 255], dtype=np.uint8)
    u_ref = np.stack([rng.normal(0, 0.5, A), rng.normal(0, 0.05, A)], axis=1)
    for dyn in (oracle.DYN_DBM, oracle.DYN_KBM, oracle.DYN_DUM):
        p = oracle.make_params(alpha=1.0, R=(0.5, 0, 0, 0.5), dynamics=dyn)
        u0, s0, r0 = oracle.solve_batch(p, state, u_ref, slots, kind, want_rows=True)
        u1, s1, r1 = oracle.solve_batch(p, state, u_ref, boxed, kind_box, want_rows=True)
        assert np.array_equal(s0, s1) and np.array_equal(u0, u1) and np.array_equal(r0, r1)
    p = oracle.make_params(alpha=1.0, R=(0.5, 0, 0, 0.5), dynamics=oracle.DYN_SADBM, flags=oracle.FLAG_CONE_BETA_STATE)
    slip0 = np.stack([rng.uniform(-0.2, 0.2, A), rng.uniform(-0.2, 0.2, A)], axis=1)
    sa, sb = slip0.copy(), slip0.copy()
    u0, s0, _ = oracle.solve_sa_batch(p, 0.05, state, sa, u_ref, slots, kind, want_rows=True)
    u1, s1, _ = oracle.solve_sa_batch(p, 0.05, state, sb, u_ref, boxed, kind_box, want_rows=True)
    assert np.array_equal(s0, s1) and np.array_equal(u0, u1) and np.array_equal(sa, sb)
