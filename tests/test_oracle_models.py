"""Rows of the unicycle-input (DUM) and steer-rate (SADBM) models in the oracle against the matrix
products the reference writes down: Gh g_c and Gh f_c with Gh = [dx, dy, dtheta, dv (, dbeta)]
(cbf/cbf.py:279-286, 398-405), g_c / f_c of cbf.py:237-245 and 338-347."""
import numpy as np
import pytest


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        s = np.array([rng.uniform(-20, 20), rng.uniform(-20, 20), rng.uniform(-np.pi, np.pi), rng.uniform(0, 15)])
        d = dict(h=rng.normal(), hx=rng.normal(), hy=rng.normal(), hth=rng.normal(), hv=rng.normal(),
                 ht=rng.normal())
        yield s, d, rng.uniform(-0.5, 0.5), rng.normal(), rng.uniform(0.2, 3.0)


def test_dum_row_is_gh_times_gc_and_fc(oracle):
    for s, d, _, _, alpha in _cases(50, 1):
        p = oracle.make_params(alpha=alpha, dynamics=oracle.DYN_DUM)
        gh = np.array([d["hx"], d["hy"], d["hth"], d["hv"]])
        g_c = np.array([[0, 0, 0, 1], [0, 0, 1, 0]], dtype=float).T        # cvxopt column lists, cbf.py:239
        f_c = np.array([s[3] * np.cos(s[2]), s[3] * np.sin(s[2]), 0.0, 0.0])
        lg, lf = gh @ g_c, gh @ f_c
        row = oracle.row(p, s, d)
        assert np.allclose(row[:2], lg, atol=1e-14)
        assert abs(row[2] + (lf + alpha * d["h"] + d["ht"])) <= 1e-13 * (1 + abs(row[2]))


def test_sadbm_row_is_gh_times_gc_and_fc(oracle):
    for s, d, beta, hb, alpha in _cases(50, 2):
        lr = 1.3
        p = oracle.make_params(alpha=alpha, lr=lr, lf=1.6, dynamics=oracle.DYN_SADBM)
        gh = np.array([d["hx"], d["hy"], d["hth"], d["hv"], hb])
        g_c = np.array([[0, 0, 0, 1, 0], [0, 0, 0, 0, 1]], dtype=float).T  # cbf.py:340
        f_c = np.array([s[3] * np.cos(s[2] + beta), s[3] * np.sin(s[2] + beta), s[3] * np.sin(beta) / lr, 0.0, 0.0])
        lg, lf = gh @ g_c, gh @ f_c
        row = oracle.row_sa(p, s, beta, d, hb)
        assert np.allclose(row[:2], lg, atol=1e-14)
        assert abs(row[2] + (lf + alpha * d["h"] + d["ht"])) <= 1e-13 * (1 + abs(row[2]))


def test_sadbm_tick_by_hand(oracle):
    """One agent, one far ellipse (no active row): the answer is the nominal input, so
    beta = beta_ref_dot dt and delta* = atan2((lf + lr) tan(beta), lr) (cbf.py:359-372, 421, 430)."""
    lr, lf, dt = 1.45, 1.45, 0.05
    p = oracle.make_params(lr=lr, lf=lf, dynamics=oracle.DYN_SADBM)
    state = np.array([[0.0, 0.0, 0.1, 5.0]])
    obs = np.zeros((1, 2, 1, 4))
    obs[0, 0, 0] = [500.0, 0.0, 3.0, 2.0]
    slip = np.array([[0.02, 0.01]])
    u, st, rows = oracle.solve_sa_batch(p, dt, state, slip, np.array([[0.7, 0.1]]), obs,
                                        np.array([oracle.OBS_ELLIPSE], dtype=np.uint8), want_rows=True)
    beta_ref = np.arctan2(lr * np.tan(0.1), lf + lr)
    beta = 0.02 + (beta_ref - 0.01) / dt * dt
    assert st[0] == 0 and rows[0, 0, 3] > 0
    assert abs(slip[0, 0] - beta) < 1e-15 and abs(slip[0, 1] - beta_ref) < 1e-15
    assert abs(u[0, 0] - 0.7) < 1e-15 and abs(u[0, 1] - np.arctan2((lf + lr) * np.tan(beta), lr)) < 1e-15


@pytest.mark.parametrize("flag", [0, 16])
def test_sadbm_cone_beta_source(oracle, flag):
    """A cone reads its own beta (slot) unless the flag hands it the agent's slip angle."""
    p = oracle.make_params(dynamics=oracle.DYN_SADBM, flags=flag)
    state = np.array([[0.0, 0.0, 0.2, 8.0]])
    slot = np.array([30.0, 2.0, 2.5, 3.0, 4.0, 0.3, 0.0, 0.0])
    obs = np.ascontiguousarray(slot.reshape(1, 2, 1, 4))
    slip = np.array([[0.05, 0.0]])
    _, _, rows = oracle.solve_sa_batch(p, 0.1, state, slip.copy(), np.array([[0.0, 0.0]]), obs,
                                       np.array([oracle.OBS_CONE], dtype=np.uint8), want_rows=True)
    o = slot.copy()
    o[5] = 0.05 if flag else 0.3
    d = oracle.barrier(oracle.OBS_CONE, state[0], o)
    exp = oracle.row_sa(p, state[0], 0.05, d, d["hth"])
    assert np.allclose(rows[0, 0, :3], exp, rtol=1e-14, atol=1e-14) and rows[0, 0, 3] == d["h"]
