"""The oracle's exact QP against closed forms, a second algorithm, SciPy and KKT (CPU only)."""
import numpy as np
import pytest
from scipy.optimize import minimize


def _params(orc, R=(1.0, 0, 0, 1.0)):
    return orc.make_params(R=R)


def _rows(arr):
    """list of (a0, a1, b) for one agent -> (M, 1, 4)"""
    r = np.zeros((len(arr), 1, 4))
    for i, (a0, a1, b) in enumerate(arr):
        r[i, 0, :3] = (a0, a1, b)
    return r


def test_no_active_row_returns_nominal(oracle):
    u, st = oracle.qp_batch(_params(oracle), _rows([(1, 0, -5), (0, 1, -2), (-1, -1, -9)]), np.array([[0.3, -0.2]]))
    assert st[0] == 0 and np.allclose(u[0], [0.3, -0.2], atol=1e-15)


@pytest.mark.parametrize("R", [np.eye(2), np.diag([0.5, 0.5]), np.array([[2.0, 0.3], [0.3, 0.7]])])
def test_single_active_row_closed_form(oracle, R):
    """u = u_nom + R^-1 a (b - a.u_nom) / (a R^-1 a^T)"""
    a, b, un = np.array([1.5, -0.7]), 2.0, np.array([0.2, 0.4])
    u, st = oracle.qp_batch(_params(oracle, R.reshape(-1)), _rows([(a[0], a[1], b), (0, 1, -50)]), un[None, :])
    Ri = np.linalg.inv(R)
    expect = un + Ri @ a * (b - a @ un) / (a @ Ri @ a)
    assert st[0] == 0 and np.allclose(u[0], expect, atol=1e-12)


def test_two_active_rows_vertex(oracle):
    rows = [(1, 0, 1.0), (0, 1, 2.0)]
    u, st = oracle.qp_batch(_params(oracle), _rows(rows), np.zeros((1, 2)))
    assert st[0] == 0 and np.allclose(u[0], [1.0, 2.0], atol=1e-12)


def test_infeasible_and_zero_rows(oracle):
    p = _params(oracle)
    u, st = oracle.qp_batch(p, _rows([(1, 0, 1.0), (-1, 0, 0.5)]), np.array([[0.1, 0.2]]))
    assert st[0] == 1 and np.allclose(u[0], [0.1, 0.2])          # pass-through
    u, st = oracle.qp_batch(p, _rows([(0, 0, 1.0)]), np.zeros((1, 2)))
    assert st[0] == 1                                             # 0.u >= 1
    u, st = oracle.qp_batch(p, _rows([(0, 0, -1.0), (1, 1, 1.0)]), np.zeros((1, 2)))
    assert st[0] == 0 and np.allclose(u[0], [0.5, 0.5])           # empty row ignored
    u, st = oracle.qp_batch(p, _rows([(np.nan, 0, 1.0)]), np.zeros((1, 2)))
    assert st[0] == 2


def _random_rows(rng, A, M, un=None):
    """Random half-planes; 85 % of the agents get rows that all contain a common point near
    u_nom (feasible by construction), the rest arbitrary offsets (often infeasible)."""
    ang = rng.uniform(0, 2 * np.pi, (M, A))
    scale = np.exp(rng.uniform(-2, 2, (M, A)))
    rows = np.zeros((M, A, 4))
    rows[..., 0] = np.cos(ang) * scale
    rows[..., 1] = np.sin(ang) * scale
    z = rng.normal(0, 1.5, (A, 2)) + (0 if un is None else un)
    az = rows[..., 0] * z[None, :, 0] + rows[..., 1] * z[None, :, 1]
    slack = np.abs(rng.normal(0, 1.0, (M, A)))
    wild = rng.uniform(size=A) < 0.15
    slack[:, wild] = rng.normal(0.3, 1.0, (M, int(wild.sum())))
    rows[..., 2] = az - slack * scale
    return rows


@pytest.mark.parametrize("M", [1, 2, 3, 8, 16, 32])
def test_enumeration_agrees_with_incremental(oracle, M):
    rng = np.random.default_rng(100 + M)
    A = 2000
    un = rng.normal(0, 1, (A, 2))
    rows = _random_rows(rng, A, M, un)
    p = _params(oracle, (0.8, 0.1, 0.1, 0.5))
    u1, s1 = oracle.qp_batch(p, rows, un)
    u2, s2 = oracle.qp_batch(p, rows, un, incremental=True)
    assert (s1 == s2).all()
    ok = s1 == 0
    assert ok.sum() > A // 2
    assert np.abs(u1[ok] - u2[ok]).max() < 1e-8


def test_degenerate_parallel_and_duplicate_rows(oracle):
    p = _params(oracle)
    rows = _rows([(1, 0, 1.0), (1, 0, 1.0), (2, 0, 2.0), (1, 1e-13, 1.0), (0, 1, -3)])
    for inc in (False, True):
        u, st = oracle.qp_batch(p, rows, np.zeros((1, 2)), incremental=inc)
        assert st[0] == 0 and np.allclose(u[0], [1.0, 0.0], atol=1e-9)


def test_kkt_residuals(oracle):
    """Stationarity, primal feasibility, complementary slackness with multipliers >= 0."""
    rng = np.random.default_rng(7)
    A, M = 500, 6
    un = rng.normal(0, 1, (A, 2))
    rows = _random_rows(rng, A, M, un)
    R = np.array([[1.3, -0.2], [-0.2, 0.6]])
    u, st = oracle.qp_batch(_params(oracle, R.reshape(-1)), rows, un)
    checked = 0
    for a in np.where(st == 0)[0]:
        Aa, b = rows[:, a, :2], rows[:, a, 2]
        slack = Aa @ u[a] - b
        assert slack.min() > -1e-9 * (1 + np.abs(b).max())
        act = np.where(slack < 1e-7 * (1 + np.abs(b)))[0]
        grad = 2 * R @ (u[a] - un[a])
        if len(act) == 0:
            assert np.abs(grad).max() < 1e-9
        else:
            lam, res, *_ = np.linalg.lstsq(Aa[act].T, grad, rcond=None)
            assert np.abs(Aa[act].T @ lam - grad).max() < 1e-6 * (1 + np.abs(grad).max())
            if len(act) <= 2:
                assert lam.min() > -1e-7 * (1 + np.abs(lam).max())
        checked += 1
    assert checked > 50


def test_against_scipy_slsqp(oracle):
    rng = np.random.default_rng(11)
    A, M = 60, 5
    un = rng.normal(0, 1, (A, 2))
    rows = _random_rows(rng, A, M, un)
    R = np.diag([0.5, 0.5])
    u, st = oracle.qp_batch(_params(oracle, R.reshape(-1)), rows, un)
    n = 0
    for a in np.where(st == 0)[0]:
        Aa, b = rows[:, a, :2], rows[:, a, 2]
        res = minimize(lambda x: (x - un[a]) @ R @ (x - un[a]), un[a], jac=lambda x: 2 * R @ (x - un[a]),
                       constraints=[{"type": "ineq", "fun": lambda x: Aa @ x - b, "jac": lambda x: Aa}],
                       method="SLSQP", options={"ftol": 1e-14, "maxiter": 200})
        if res.success:
            assert np.abs(res.x - u[a]).max() < 1e-5 * (1 + np.abs(u[a]).max()), (a, res.x, u[a])
            n += 1
    assert n > 20


def _interior_point(R, u0, G, b, abstol=1e-7, reltol=1e-6, feastol=1e-7, maxiters=100):
    """Primal-dual path-following method with Mehrotra's predictor-corrector for
        min (u - u0)' R (u - u0)   s.t.   G u >= b,
    the class of algorithm the reference's solver runs (cvxopt.solvers.cp, reference cbf/cbf.py:213: Newton steps
    on the perturbed KKT system, stopping on the duality gap and the residuals), with cvxopt's default
    tolerances (abstol 1e-7, reltol 1e-6, feastol 1e-7, 100 iterations).  cvxopt is not installed here and the
    reference pins no version: this is a restatement of the published method, not of its code.
    Returns u, or None when the iteration does not converge (an infeasible problem)."""
    m = len(b)
    P, q = 2.0 * R, -2.0 * R @ u0
    u, s, z = u0.copy(), np.ones(m), np.ones(m)
    s = np.maximum(G @ u - b, 1.0)
    for _ in range(maxiters):
        rd = P @ u + q - G.T @ z            # stationarity
        rp = G @ u - s - b                  # primal residual
        gap = s @ z
        f0 = 0.5 * u @ P @ u + q @ u
        if (np.linalg.norm(rp) <= feastol * max(1.0, np.linalg.norm(b)) and
                np.linalg.norm(rd) <= feastol * max(1.0, np.linalg.norm(q)) and
                (gap <= abstol or gap <= reltol * abs(f0))):
            return u

        def newton(rc):
            # (P + G' diag(z/s) G) du = -rd + G' ((rc - z rp) / s) ... eliminated from the 3 x 3 block system
            w = z / s
            H = P + G.T @ (w[:, None] * G)
            du = np.linalg.solve(H, -rd + G.T @ ((rc - z * rp) / s))
            ds = G @ du + rp
            dz = (rc - z * ds) / s
            return du, ds, dz

        def step(v, dv):
            neg = dv < 0
            return min(1.0, 0.99 * np.min(-v[neg] / dv[neg])) if neg.any() else 1.0

        du, ds, dz = newton(-s * z)                                  # predictor
        a = min(step(s, ds), step(z, dz))
        sigma = (((s + a * ds) @ (z + a * dz)) / gap) ** 3
        du, ds, dz = newton(-s * z - ds * dz + sigma * gap / m)      # corrector
        a = min(step(s, ds), step(z, dz))
        u, s, z = u + a * du, s + a * ds, z + a * dz
    return None


def test_interior_point_at_the_reference_solvers_tolerances_lands_on_the_oracle(oracle):
    """What an interior-point solver stopped at cvxopt's default tolerances returns is the oracle's exact optimum to
    those tolerances: multi-row problems, for which the reference holds no numbers, 1 ... 16 rows."""
    rng = np.random.default_rng(2026)
    R = np.array([[0.5, 0.0], [0.0, 0.5]])          # the drivers' weight, stanley_controller_ellipse.py:741
    n, worst = 0, 0.0
    for M in (1, 2, 3, 5, 8, 16):
        A = 60
        un = rng.normal(0, 1, (A, 2))
        rows = _random_rows(rng, A, M, un)
        u, st = oracle.qp_batch(_params(oracle, R.reshape(-1)), rows, un)
        for a in np.where(st == 0)[0]:
            x = _interior_point(R, un[a], rows[:, a, :2], rows[:, a, 2])
            assert x is not None, (M, a)
            err = np.abs(x - u[a]).max() / (1 + np.abs(u[a]).max())
            worst = max(worst, err)
            n += 1
    assert n > 250
    # measured 7.9e-5: the stopping rule (duality gap 1e-7, or 1e-6 of the cost) leaves the iterate that far from the
    # exact optimum; it is the accuracy to which any number the reference produced was ever defined
    assert worst <= 2e-4, worst


def test_least_violation_answer_for_infeasible_rows(oracle):
    """Rows without a common point: the oracle's answer minimises the largest violation (distance to the row's
    half-plane in the metric of the cost) and, among such inputs, is the one nearest to the nominal input.  Checked
    against a linear program for the least worst-case violation (scipy HiGHS) and SLSQP for the nearest point."""
    from scipy.optimize import linprog
    rng = np.random.default_rng(77)
    R = np.array([[2.0, 0.3], [0.3, 1.0]])
    Lc = np.linalg.cholesky(R)
    p = oracle.make_params(R=R.reshape(-1))
    A, M = 400, 6
    ang = rng.uniform(-np.pi, np.pi, (M, A))
    rows = np.zeros((M, A, 4))
    rows[:, :, 0], rows[:, :, 1] = np.cos(ang) * rng.uniform(0.1, 10, (M, A)), np.sin(ang) * rng.uniform(0.1, 10, (M, A))
    rows[:, :, 2] = rng.normal(3.0, 3.0, (M, A))
    rows[M - 1, ::7, :3] = (0.0, 0.0, -1.0)            # an empty row among them
    un = rng.normal(0, 1, (A, 2))
    u0, st0 = oracle.qp_batch(p, rows, un)
    u, st, t = oracle.qp_min_violation_batch(p, rows, un)
    assert np.array_equal(st, st0) and (st == 1).sum() > 150 and (st == 0).sum() > 50
    assert np.array_equal(u[st == 0], u0[st == 0]) and (t[st == 0] == 0).all()
    checked = 0
    for a in np.where(st == 1)[0][:120]:
        a_u, b_u = rows[:, a, :2], rows[:, a, 2]
        keep = np.hypot(a_u[:, 0], a_u[:, 1]) > 0
        at = np.linalg.solve(Lc, a_u[keep].T).T                    # normals in w = Lc^T (u - u_nom)
        nrm = np.hypot(at[:, 0], at[:, 1])
        at, bt = at / nrm[:, None], (b_u[keep] - a_u[keep] @ un[a]) / nrm
        res = linprog(c=[0, 0, 1.0], A_ub=np.c_[-at, -np.ones(len(bt))], b_ub=-bt, bounds=[(None, None)] * 3, method="highs")
        assert res.status == 0 and abs(res.x[2] - t[a]) <= 1e-8 * (1 + abs(t[a])), (a, res.x[2], t[a])
        w = Lc.T @ (u[a] - un[a])
        viol = (bt - at @ w).max()
        assert abs(viol - t[a]) <= 1e-7 * (1 + abs(t[a]))          # the answer attains the least worst-case violation
        # and it is the nearest such point to the nominal input: the inputs that attain t* (here to 1e-7) form a set whose
        # bounding box four linear programs give; generically that set is one point, and the answer lies in it
        box = []
        for cvec in ([1, 0], [-1, 0], [0, 1], [0, -1]):
            rb = linprog(c=cvec, A_ub=-at, b_ub=-(bt - t[a] * (1 + 1e-7) - 1e-9), bounds=[(None, None)] * 2, method="highs")
            box.append(rb.x if rb.status == 0 else None)
        if any(b is None for b in box):
            continue
        lo = np.array([box[0][0], box[2][1]])
        hi = np.array([box[1][0], box[3][1]])
        assert (w >= lo - 1e-6 * (1 + np.abs(w))).all() and (w <= hi + 1e-6 * (1 + np.abs(w))).all(), (a, w, lo, hi)
        if np.abs(hi - lo).max() > 1e-4 * (1 + np.abs(w).max()):
            continue                                                # a segment of equally good inputs: not judged here
        checked += 1
    assert checked >= 100
    # hand-made: two rows facing each other with a gap of 1.5 (R = I): the answer is the midline point nearest to u_nom
    p1 = oracle.make_params(R=(1, 0, 0, 1))
    r = np.zeros((2, 1, 4))
    r[0, 0, :3] = (1, 0, 1.0)
    r[1, 0, :3] = (-1, 0, 0.5)
    u, st, t = oracle.qp_min_violation_batch(p1, r, np.array([[0.3, 0.7]]))
    assert st[0] == 1 and abs(t[0] - 0.75) < 1e-12 and np.allclose(u[0], [0.25, 0.7], atol=1e-9)
    # the empty triangle of the hand-made cases: x >= 1, y >= 2, x + y <= 2.9
    r = np.zeros((3, 1, 4))
    r[0, 0, :3], r[1, 0, :3], r[2, 0, :3] = (1, 0, 1.0), (0, 1, 2.0), (-1, -1, -2.9)
    u, st, t = oracle.qp_min_violation_batch(p1, r, np.array([[0.0, 0.0]]))
    tt = 0.1 / (2 + np.sqrt(2))                                    # equal distance to the three lines
    assert st[0] == 1 and abs(t[0] - tt) < 1e-12 and np.allclose(u[0], [1 - tt, 2 - tt], atol=1e-9)
