"""Barrier formulas of the oracle: finite differences where the reference's partials are
true derivatives, and the second textual statement of each row in the reference's driver
script (inline CBF_A / CBF_cone / CBF_lane, test_scripts/stanley_controller_ellipse.py)
restated here in numpy as an independent check (CPU only)."""
import numpy as np
import pytest
from scipy.optimize import minimize

EPS = 1e-3
LR = 1.45


def _fd(oracle, kind, s, o, i, h=1e-6):
    sp, sm = np.array(s, float), np.array(s, float)
    sp[i] += h
    sm[i] -= h
    return (oracle.barrier(kind, sp, o)["h"] - oracle.barrier(kind, sm, o)["h"]) / (2 * h)


def test_ellipse_partials_are_derivatives(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        s = [rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-3, 3), rng.uniform(0, 15)]
        o = [rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(2, 20), rng.uniform(1, 10),
             rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-3, 3), 0]
        d = oracle.barrier(oracle.OBS_ELLIPSE, s, o)
        assert abs(d["hx"] - _fd(oracle, 0, s, o, 0)) < 1e-6 * (1 + abs(d["hx"]))
        assert abs(d["hy"] - _fd(oracle, 0, s, o, 1)) < 1e-6 * (1 + abs(d["hy"]))
        assert d["hth"] == 0 and d["hv"] == 0
        # inline statement CBF_A (stanley_controller_ellipse.py:290-292) for theta_o = 0
        o0 = list(o)
        o0[4] = 0.0
        d0 = oracle.barrier(oracle.OBS_ELLIPSE, s, o0)
        xd, yd = s[0] - o[0], s[1] - o[1]
        assert abs(d0["h"] - ((xd / o[2]) ** 2 + (yd / o[3]) ** 2 - 1)) < 1e-12
        assert abs(d0["hx"] - 2 * xd / o[2] ** 2) < 1e-12 and abs(d0["hy"] - 2 * yd / o[3] ** 2) < 1e-12
        # dt as the reference writes it (no rotation): -2 (xd vx / a^2 + yd vy / b^2)
        assert abs(d["ht"] + 2 * (xd * o[5] / o[2] ** 2 + yd * o[6] / o[3] ** 2)) < 1e-12


def test_radial_matches_inline_statement(oracle):
    """radial_dynamic_obstacles.py:391-405"""
    rng = np.random.default_rng(1)
    for _ in range(50):
        s = [rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-3, 3), rng.uniform(0, 10)]
        c = rng.uniform(-20, 20, 2)
        a = b = rng.uniform(1.5, 2.0)
        cd = rng.uniform(-5, 5, 2)
        kv = 1.0
        d = oracle.barrier(oracle.OBS_RADIAL, s, [c[0], c[1], a, b, cd[0], cd[1], kv, 0])
        h1 = ((s[0] - c[0]) / a) ** 2 + ((s[1] - c[1]) / b) ** 2 - 1 - (kv * s[3] / (1 + s[3]))
        assert abs(d["h"] - h1) < 1e-12
        assert abs(d["hx"] - 2 * (s[0] - c[0]) / a ** 2) < 1e-12
        assert abs(d["hy"] - 2 * (s[1] - c[1]) / b ** 2) < 1e-12
        assert abs(d["hv"] + kv / (1 + s[3]) ** 2) < 1e-12
        assert abs(d["ht"] + 2 * (((s[0] - c[0]) / a ** 2) * cd[0] + ((s[1] - c[1]) / b ** 2) * cd[1])) < 1e-12
        assert abs(d["hv"] - _fd(oracle, 3, s, [c[0], c[1], a, b, cd[0], cd[1], kv, 0], 3)) < 1e-6


def _inline_cone(s, s_obs, cone_a):
    """CBF_cone of stanley_controller_ellipse.py:331-388 (no ZERO_TOL regularisation)."""
    cx, cy = s_obs[0], s_obs[1]
    p_rel = np.array([s[0] - cx, s[1] - cy])
    dist = np.linalg.norm(p_rel)
    svx, svy = s[3] * np.cos(s[2]), s[3] * np.sin(s[2])
    ovx, ovy = s_obs[3] * np.cos(s_obs[2]), s_obs[3] * np.sin(s_obs[2])
    v_rel = np.array([svx - ovx, svy - ovy])
    cb = np.sqrt(dist ** 2 - cone_a ** 2)
    vn = np.linalg.norm(v_rel)
    out = {}
    out["h"] = p_rel @ v_rel + dist * vn * cb / dist
    out["hx"] = v_rel[0] + vn * p_rel[0] / cb
    out["hy"] = v_rel[1] + vn * p_rel[1] / cb
    out["hth"] = -p_rel[0] * svy + p_rel[1] * svx + (-v_rel[0] * svy + v_rel[1] * svx) * cb / vn
    out["hv"] = (p_rel[0] * np.cos(s[2]) + p_rel[1] * np.sin(s[2])
                 + (v_rel[0] * np.cos(s[2]) + v_rel[1] * np.sin(s[2])) * cb / vn)
    out["ht"] = -v_rel[0] * ovx - v_rel[1] * ovy - vn * (p_rel[0] * ovx + p_rel[1] * ovy) / cb
    return out


def test_cone_matches_inline_statement_up_to_regularisation(oracle):
    """The class adds ZERO_TOL in three places (cbf/obstacles.py:428,442,496); away from the
    degenerate set the inline statement agrees to O(ZERO_TOL / scale)."""
    rng = np.random.default_rng(2)
    for _ in range(100):
        s = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-3, 3), rng.uniform(2, 15)])
        br, rg = rng.uniform(0, 2 * np.pi), rng.uniform(15, 60)
        so = np.array([s[0] + rg * np.cos(br), s[1] + rg * np.sin(br), rng.uniform(-3, 3), rng.uniform(0, 1.5)])
        a = rng.uniform(2, 6)
        d = oracle.barrier(oracle.OBS_CONE, s, [so[0], so[1], so[2], so[3], a, 0.0, 0, 0])
        e = _inline_cone(s, so, a)
        scale = 1 + rg * (s[3] + so[3])      # size of the terms that cancel inside each partial
        for k in ("h", "hx", "hy", "hth", "hv", "ht"):
            assert abs(d[k] - e[k]) < 2e-3 * scale, (k, d[k], e[k])


def test_cone_degenerate_branches(oracle):
    """inside the radius: cone_boundary = ZERO_TOL; at the centre: cos_phi = 0."""
    s = [0.0, 0.0, 0.3, 5.0]
    d = oracle.barrier(oracle.OBS_CONE, s, [1.0, 0.0, 0.0, 0.0, 3.0, 0.0, 0, 0])   # dist 1 < a 3
    vx, vy = 5 * np.cos(0.3), 5 * np.sin(0.3)
    assert abs(d["h"] - ((-1.0) * vx + 1.0 * 5.0 * (EPS / 1.0))) < 1e-12
    assert abs(d["hx"] - (vx + 5.0 * (-1.0) / (2 * EPS))) < 1e-9
    d = oracle.barrier(oracle.OBS_CONE, s, [0.0, 0.0, 0.0, 0.0, 3.0, 0.0, 0, 0])   # dist 0
    assert abs(d["h"]) < 1e-12
    # ego and obstacle with identical velocity: v_rel = 0, partials stay finite
    d = oracle.barrier(oracle.OBS_CONE, [0, 0, 0.0, 2.0], [10.0, 0.0, 0.0, 2.0, 3.0, 0.0, 0, 0])
    assert all(np.isfinite(v) for v in d.values())


def _stationary_minima(c, px, py):
    """all local minimisers of (x-px)^2 + (p(x)-py)^2 via the roots of its derivative"""
    P = np.polynomial.Polynomial(c)
    g = (np.polynomial.Polynomial([-px, 1.0])) ** 2 + (P - py) ** 2
    r = g.deriv().roots()
    r = r[np.abs(r.imag) < 1e-9].real
    return [x for x in r if g.deriv(2)(x) > 0]


def test_lane_closest_point_lines_closed_form(oracle):
    rng = np.random.default_rng(3)
    for _ in range(100):
        c0, c1 = rng.uniform(-20, 20), rng.uniform(-2, 2)
        px, py = rng.uniform(-90, 10), rng.uniform(-30, 30)
        x = oracle.lane_closest_x([c0, c1], px, py)
        assert abs(x - (px + c1 * (py - c0)) / (1 + c1 * c1)) < 1e-10 * (1 + abs(x))


def test_lane_closest_point_cubics_vs_scipy_newton_cg(oracle):
    """Where the distance has one local minimum, the oracle, the polynomial roots and scipy's
    Newton-CG from x0 = px (the reference's solver call, obstacles.py:675-677, given the
    true Hessian) must agree."""
    rng = np.random.default_rng(4)
    checked = 0
    for _ in range(300):
        c = [rng.uniform(-5, 5), rng.uniform(-1, 1), rng.uniform(-0.05, 0.05), rng.uniform(-0.002, 0.002)]
        px, py = rng.uniform(-20, 20), rng.uniform(-20, 20)
        mins = _stationary_minima(c, px, py)
        x = oracle.lane_closest_x(c, px, py)
        # always a local minimum
        assert min(abs(x - m) for m in mins) < 1e-6 * (1 + abs(x))
        if len(mins) != 1:
            continue
        P = np.polynomial.Polynomial(c)
        g = lambda z: (z[0] - px) ** 2 + (P(z[0]) - py) ** 2
        dg = lambda z: np.array([2 * (z[0] - px) + 2 * (P(z[0]) - py) * P.deriv()(z[0])])
        ddg = lambda z: np.array([[2 * (1 + P.deriv()(z[0]) ** 2 + (P(z[0]) - py) * P.deriv(2)(z[0]))]])
        res = minimize(g, np.array([px]), method="Newton-CG", jac=dg, hess=ddg, options={"xtol": 1e-8})
        assert abs(res.x[0] - x) < 1e-5 * (1 + abs(x))
        checked += 1
    assert checked > 100


def test_lane_of_degree_up_to_six(oracle):
    """PolyLane takes a coefficient array of any length (cbf/obstacles.py:578-593, numpy Polynomial and its
    derivatives): kind 8 holds seven coefficients.  The closest point is a local minimum of the distance and scipy's
    Newton-CG from x0 = px (the reference's call with the true Hessian) finds the same one where there is only one;
    the row is the reference's formula on numpy's Polynomial; a cubic gives what the cubic kind gives."""
    rng = np.random.default_rng(41)
    checked = 0
    for i in range(200):
        deg = 4 + i % 3
        c = np.zeros(7)
        c[:deg + 1] = [rng.uniform(-5, 5), rng.uniform(-0.5, 0.5), rng.uniform(-0.02, 0.02), rng.uniform(-5e-4, 5e-4),
                       rng.uniform(-1e-5, 1e-5), rng.uniform(-2e-7, 2e-7), rng.uniform(-3e-9, 3e-9)][:deg + 1]
        px, py = rng.uniform(-20, 20), rng.uniform(-20, 20)
        P = np.polynomial.Polynomial(c)
        x = oracle.lane_closest_x(c, px, py)
        d1 = 2 * (x - px) + 2 * (P(x) - py) * P.deriv()(x)
        d2 = 2 * (1 + P.deriv()(x) ** 2 + (P(x) - py) * P.deriv(2)(x))
        assert abs(d1) <= 1e-6 * (1 + abs(d2)) and d2 > 0                      # a local minimum of the distance
        s = [px, py, 0.3, 4.0]
        buf = 0.9
        d = oracle.barrier(oracle.OBS_LANE6, s, list(c) + [buf])
        g, dg, ddg = P(x), P.deriv()(x), P.deriv(2)(x)
        eta = 1 + dg * ddg + dg ** 2 - py * ddg
        if abs(eta) < EPS:
            eta = EPS
        assert abs(d["h"] - ((x - px) ** 2 + (g - py) ** 2 - buf)) < 1e-9
        assert abs(d["hx"] - (2 / eta) * ((px - x) * (eta - 1) - (py - g) * dg)) < 1e-8 * (1 + abs(d["hx"]))
        assert abs(d["hy"] - (2 / eta) * (-(px - x) * dg + (py - g) * (eta - dg ** 2))) < 1e-8 * (1 + abs(d["hy"]))
        assert d["hth"] == 0 and d["hv"] == 0 and d["ht"] == 0
        mins = _stationary_minima(c, px, py)
        if len(mins) == 1:
            dist = lambda z: (z[0] - px) ** 2 + (P(z[0]) - py) ** 2                                   # noqa: E731
            jac = lambda z: np.array([2 * (z[0] - px) + 2 * (P(z[0]) - py) * P.deriv()(z[0])])        # noqa: E731
            hes = lambda z: np.array([[2 * (1 + P.deriv()(z[0]) ** 2 + (P(z[0]) - py) * P.deriv(2)(z[0]))]])   # noqa: E731
            res = minimize(dist, np.array([px]), method="Newton-CG", jac=jac, hess=hes, options={"xtol": 1e-8})
            assert abs(res.x[0] - x) < 1e-5 * (1 + abs(x))
            checked += 1
    assert checked > 60
    for _ in range(30):      # a cubic through both kinds
        c = [rng.uniform(-5, 5), rng.uniform(-1, 1), rng.uniform(-0.05, 0.05), rng.uniform(-0.002, 0.002)]
        s = [rng.uniform(-20, 20), rng.uniform(-20, 20), 0.1, 3.0]
        a = oracle.barrier(oracle.OBS_LANE, s, c + [1.5, 0, 0, 0])
        b = oracle.barrier(oracle.OBS_LANE6, s, c + [0, 0, 0, 1.5])
        for k in ("h", "hx", "hy"):
            assert abs(a[k] - b[k]) <= 1e-12 * (1 + abs(a[k])), k


def test_lane_row_matches_inline_statement(oracle):
    """CBF_lane, stanley_controller_ellipse.py:440-446, with xc from the oracle."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        c = [rng.uniform(-5, 5), rng.uniform(-1, 1), rng.uniform(-0.05, 0.05), rng.uniform(-0.002, 0.002)]
        s = [rng.uniform(-20, 20), rng.uniform(-20, 20), 0.1, 3.0]
        buf = 1.5
        d = oracle.barrier(oracle.OBS_LANE, s, c + [buf, 0, 0, 0])
        xc = oracle.lane_closest_x(c, s[0], s[1])
        P = np.polynomial.Polynomial(c)
        g, dg, ddg = P(xc), P.deriv()(xc), P.deriv(2)(xc)
        eta = 1 + dg * ddg + dg ** 2 - s[1] * ddg
        if abs(eta) < EPS:
            eta = EPS
        assert abs(d["h"] - ((xc - s[0]) ** 2 + (g - s[1]) ** 2 - buf)) < 1e-10
        assert abs(d["hx"] - (2 / eta) * ((s[0] - xc) * (eta - 1) - (s[1] - g) * dg)) < 1e-9
        assert abs(d["hy"] - (2 / eta) * (-(s[0] - xc) * dg + (s[1] - g) * (eta - dg ** 2))) < 1e-9


@pytest.mark.parametrize("kind", ["ellipse", "cone"])
def test_dbm_row_matches_inline_contraction(oracle, kind):
    """Row assembly: Dh1 (f_c + g_c x) + alpha h + h_t >= 0 with the f_c, g_c of
    stanley_controller_ellipse.py:295-298 / 391-394 (cvxopt column-major g_c)."""
    rng = np.random.default_rng(6)
    p = oracle.make_params(alpha=0.7, lr=LR, lf=LR)
    for _ in range(30):
        s = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-3, 3), rng.uniform(1, 10)])
        if kind == "ellipse":
            o = [20.0, 3.0, 6.0, 3.0, 0.4, 1.0, -0.5, 0]
            d = oracle.barrier(oracle.OBS_ELLIPSE, s, o)
        else:
            o = [25.0, 4.0, 0.3, 2.0, 4.0, 0.0, 0, 0]
            d = oracle.barrier(oracle.OBS_CONE, s, o)
        row = oracle.row(p, s, d)
        Dh = np.array([d["hx"], d["hy"], d["hth"], d["hv"]])
        f_c = np.array([s[3] * np.cos(s[2]), s[3] * np.sin(s[2]), 0, 0])
        g_c = np.array([[0, 0, 0, 1], [-s[3] * np.sin(s[2]), s[3] * np.cos(s[2]), s[3] / LR, 0]]).T  # 4x2
        assert np.allclose(row[:2], Dh @ g_c, atol=1e-12)
        assert abs(row[2] + (Dh @ f_c + 0.7 * d["h"] + d["ht"])) < 1e-10


def test_kbm_row(oracle):
    """KBM_VC_CBF2D: Gh g_c with g_c = [[cos, sin, 0], [0, 0, 1]] columns, b = -alpha h
    (cbf/cbf.py:94-101)."""
    p = oracle.make_params(alpha=2.0, dynamics=oracle.DYN_KBM)
    s = np.array([1.0, 2.0, 0.6, 0.0])
    d = oracle.barrier(oracle.OBS_ELLIPSE, s, [10.0, 3.0, 4.0, 2.0, 0.0, 0, 0, 0])
    row = oracle.row(p, s, d)
    assert abs(row[0] - (np.cos(0.6) * d["hx"] + np.sin(0.6) * d["hy"])) < 1e-12
    assert row[1] == d["hth"] and abs(row[2] + 2.0 * d["h"]) < 1e-12


def test_seeker_step_known_answer(oracle):
    """radial_dynamic_obstacles.py:210-234 by hand: obstacle at (3, 4), ego at the origin: distance 5,
    speed max(0.2 * 5, 3) = 3 toward the ego, 0.1 s; a far one at (-30, 40) moves at 0.2 * 50 = 10."""
    obs = np.zeros((3, 2, 2, 4))
    obs[0, 0, :, 0:2] = [[3.0, 4.0], [-30.0, 40.0]]      # radial
    obs[1, 0, :, 0:2] = [[3.0, 4.0], [-30.0, 40.0]]      # ellipse: theta_o follows the yaw
    obs[2, 0, :, :] = [[1.0, 2.0, np.pi / 2, 5.0]] * 2   # cone: constant velocity along theta_o + beta
    ty = np.array([oracle.OBS_RADIAL, oracle.OBS_ELLIPSE, oracle.OBS_CONE], dtype=np.uint8)
    oracle.seeker_step(obs, ty, np.zeros((2, 2)), 0.1)
    for m in (0, 1):
        assert np.allclose(obs[m, 0, 0, 0:2], [3.0 - 0.18, 4.0 - 0.24], atol=1e-15)
        assert np.allclose(obs[m, 0, 1, 0:2], [-30.0 + 0.6, 40.0 - 0.8], atol=1e-14)
    assert np.allclose(obs[0, 1, 0, 0:2], [-1.8, -2.4], atol=1e-15)                    # cdot
    assert np.allclose(obs[1, 1, 0, 0:3], [np.arctan2(-4.0, -3.0), -1.8, -2.4], atol=1e-15)
    assert np.allclose(obs[2, 0, 0, 0:2], [1.0, 2.5], atol=1e-15)


def test_nominal_of_a_diverged_state_returns(oracle):
    """A heading of 1e17 rad or infinity (a rollout that diverged) must not keep the angle wrap busy."""
    traj = np.array([[0.0, 0.0, 0.1, 5.0], [1.0, 0.0, 0.1, 5.0]])
    state = np.array([[0.0, 1.0, 1e17, 3.0], [0.0, 1.0, np.inf, 3.0], [0.0, 1.0, 7.0, 3.0]])
    u = oracle.nominal_batch(state, traj, np.zeros(3, dtype=np.int32))
    assert np.isfinite(u[0]).all() and abs(u[0, 1]) <= np.pi + np.pi / 2 + 1e-9
    assert not np.isfinite(u[1, 1])
    assert np.isfinite(u[2]).all()
