"""SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION: for rows that admit no input the filter hands back the input whose largest row
violation is least (the nearest such input to the nominal one) instead of the nominal input.  Against the oracle's exact
enumeration over row pairs and triples.  Needs an MI355X."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
R2 = np.array([[2.0, 0.3], [0.3, 1.0]])


def _wspace(rows, un, R):
    """Unit normals and right-hand sides in w = Lc^T (u - u_nom): (M,A,2), (M,A); zero rows flagged."""
    Lc = np.linalg.cholesky(R)
    at = np.linalg.solve(Lc, rows[:, :, :2].reshape(-1, 2).T).T.reshape(rows.shape[0], rows.shape[1], 2)
    nrm = np.hypot(at[..., 0], at[..., 1])
    zero = nrm == 0
    nrm = np.where(zero, 1.0, nrm)
    bt = (rows[:, :, 2] - np.einsum("mak,ak->ma", rows[:, :, :2], un)) / nrm
    return at / nrm[..., None], bt, zero, Lc


def _judge(ug, uo, so, t, rows, un, R, dtype, frac_close=0.97):
    """ug: the GPU's inputs, uo / so / t: the oracle's.  Infeasible agents: the GPU's answer attains the least
    worst-case violation (to the precision's tolerance), is not farther from the nominal input than the oracle's, and
    equals it wherever the set of such inputs is a point met at decent angles."""
    inf = so == 1
    at, bt, zero, Lc = _wspace(rows, un, R)
    w_g = (ug - un) @ Lc
    w_o = (uo - un) @ Lc
    viol = np.where(zero, -np.inf, bt - np.einsum("mak,ak->ma", at, w_g)).max(axis=0)
    rel = 1e-9 if dtype == torch.float64 else 2e-3     # fp32: the rows themselves differ from the oracle's by 1e-4 and more
    scale = 1 + np.abs(t) + np.abs(w_o).sum(axis=1)
    assert (viol[inf] <= t[inf] + rel * scale[inf]).all(), np.max((viol - t)[inf] / scale[inf])
    pos = inf & (t > 0)       # (t = 0: only rows that no input satisfies stood in the way, the others may hold with a margin)
    assert (viol[pos] >= t[pos] - rel * scale[pos]).all()
    n_g, n_o = np.hypot(w_g[:, 0], w_g[:, 1]), np.hypot(w_o[:, 0], w_o[:, 1])
    tol = 1e-6 if dtype == torch.float64 else 2e-3
    du = np.abs(ug - uo).max(axis=1) / np.maximum(1.0, np.abs(uo).max(axis=1))
    close = du[inf] <= tol
    assert close.mean() >= frac_close, close.mean()
    assert (n_g[inf] <= n_o[inf] * (1 + 10 * tol) + 10 * tol * scale[inf])[close].all()
    return int(inf.sum())


def _random_rows(rng, A, M, shift=3.0):
    ang = rng.uniform(-np.pi, np.pi, (M, A))
    rows = np.zeros((M, A, 4))
    rows[:, :, 0], rows[:, :, 1] = np.cos(ang) * rng.uniform(0.1, 10, (M, A)), np.sin(ang) * rng.uniform(0.1, 10, (M, A))
    rows[:, :, 2] = rng.normal(shift, 3.0, (M, A))
    return rows


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("M", [3, 6, 16, 40])
def test_qp_stage_least_violation(oracle, dtype, M):
    rng = np.random.default_rng(500 + M)
    A = 3000
    rows = _random_rows(rng, A, M, shift=3.0 if M < 16 else 1.0)
    rows[M - 1, ::9, :3] = (0.0, 0.0, -1.0)                  # empty rows among them
    un = rng.normal(0, 1, (A, 2))
    p = oracle.make_params(R=R2.reshape(-1))
    uo, so, t = oracle.qp_min_violation_batch(p, rows, un)
    u_plain, s_plain = oracle.qp_batch(p, rows, un)
    flt = BatchedCBF(R=R2, min_violation=True)
    ug, sg = flt.solve_qp(torch.from_numpy(rows).to(DEV).to(dtype), torch.from_numpy(un).to(DEV).to(dtype))
    torch.cuda.synchronize()
    ug, sg = ug.double().cpu().numpy(), sg.cpu().numpy()
    agree = sg == so
    assert agree.mean() >= (1.0 if dtype == torch.float64 else 0.999)
    feas = agree & (so == 0)
    tolf = 1e-8 if dtype == torch.float64 else 2e-3
    if feas.any() and dtype == torch.float64:      # the agents with admissible inputs: the ordinary optimum, as without the flag
        assert (np.abs(ug - uo)[feas].max(axis=1) / np.maximum(1.0, np.abs(uo[feas]).max(axis=1))).max() <= tolf
    n = _judge(ug[agree], uo[agree], so[agree], t[agree], rows[:, agree], un[agree], R2, dtype)
    assert n > 300
    moved = np.abs(ug - un).max(axis=1) > 1e-6
    assert moved[agree & (so == 1)].mean() > 0.95            # no longer the nominal input
    # without the flag: the nominal input, as before
    u0, s0 = BatchedCBF(R=R2).solve_qp(torch.from_numpy(rows).to(DEV).to(dtype), torch.from_numpy(un).to(DEV).to(dtype))
    s0 = s0.cpu().numpy()
    assert np.array_equal(u0.double().cpu().numpy()[s0 == 1], un.astype(np.float32 if dtype == torch.float32 else np.float64).astype(np.float64)[s0 == 1])


def test_frozen_infeasible_cases(oracle):
    """The infeasible QPs of tests/golden/oracle_cases.npz (hand-made and random), fp64."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cases.npz"))
    total = 0
    for key, R in (("qp_hand", np.eye(2)), ("qp_rand", np.array([[2.0, 0.3], [0.3, 1.0]]))):
        rows, un, st = g[key + "_rows"], g[key + "_u_nom"], g[key + "_status"]
        p = oracle.make_params(R=R.reshape(-1))
        uo, so, t = oracle.qp_min_violation_batch(p, rows, un)
        assert np.array_equal(so, st)
        flt = BatchedCBF(R=R, min_violation=True)
        ug, sg = flt.solve_qp(torch.from_numpy(rows).to(DEV), torch.from_numpy(un).to(DEV))
        torch.cuda.synchronize()
        ug, sg = ug.cpu().numpy(), sg.cpu().numpy()
        assert np.array_equal(sg, st)
        assert np.array_equal(ug[st == 0], g[key + "_u"][st == 0]) or np.allclose(ug[st == 0], g[key + "_u"][st == 0], rtol=1e-9, atol=1e-9)
        imp = (rows[:, :, 0] == 0) & (rows[:, :, 1] == 0) & (rows[:, :, 2] > 0)     # 0 . u >= 1: nothing to minimise
        judged = ~imp.any(axis=0)
        total += _judge(ug[judged], uo[judged], so[judged], t[judged], rows[:, judged], un[judged], R, torch.float64,
                        frac_close=0.95)
    assert total >= 85


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("dynamics,slip_io", [("dbm", True), ("dbm", False), ("kbm", False), ("dum", False)])
def test_filter_with_least_violation(oracle, dtype, dynamics, slip_io):
    """The whole tick on the bench's scene (30 % of the agents have no admissible input): rows by the filter, second
    pass on them; compared in QP coordinates with the oracle's enumeration on the oracle's rows."""
    A = 4096
    w = scenarios.config4(A=A, nominal="uniform")
    if dynamics == "kbm":
        w["u_ref"][:, 0] = np.abs(w["u_ref"][:, 0]) + 1.0
    R = np.diag([0.5, 0.5])
    dyn = {"dbm": oracle.DYN_DBM, "kbm": oracle.DYN_KBM, "dum": oracle.DYN_DUM}[dynamics]
    # the oracle in QP coordinates: rows and the nominal input as the QP sees them
    pq = oracle.make_params(alpha=1.0, R=R.reshape(-1), dynamics=dyn, flags=oracle.FLAG_SLIP_IO if slip_io else 0)
    _, so, rows = oracle.solve_batch(pq, w["state"], w["u_ref"], w["obs"], w["obs_type"], want_rows=True)
    un = w["u_ref"].copy()
    if dynamics == "dbm" and not slip_io:
        un[:, 1] = np.arctan2(1.45 * np.tan(un[:, 1]), 2.9)
    elif dynamics == "kbm":
        un[:, 1] = un[:, 0] * np.tan(un[:, 1]) / 2.9
    uo, so2, t = oracle.qp_min_violation_batch(oracle.make_params(R=R.reshape(-1)), rows, un)
    assert np.array_equal(so2[so != 2], so[so != 2])
    flt = BatchedCBF(alpha=1.0, R=R, dynamics=dynamics, slip_io=slip_io, min_violation=True)
    t_ = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in w.items()}
    ug, sg, rg = flt.solve(t_["state"].to(dtype), t_["u_ref"].to(dtype), t_["obs"].to(dtype).contiguous(), t_["obs_type"])
    plain = BatchedCBF(alpha=1.0, R=R, dynamics=dynamics, slip_io=slip_io)
    up, sp = plain.solve(t_["state"].to(dtype), t_["u_ref"].to(dtype), t_["obs"].to(dtype).contiguous(), t_["obs_type"])
    torch.cuda.synchronize()
    assert torch.equal(sg, sp)                                        # the verdicts are the filter's
    assert torch.equal(ug[sp == 0], up[sp == 0])                      # and so are the answers of everybody else
    ug, sg = ug.double().cpu().numpy(), sg.cpu().numpy()
    # back to QP coordinates
    if dynamics == "dbm" and not slip_io:
        ug[:, 1] = np.arctan2(1.45 * np.tan(ug[:, 1]), 2.9)
    elif dynamics == "kbm":
        ug[:, 1] = w["u_ref"][:, 0] * np.tan(ug[:, 1]) / 2.9
    agree = (sg == so) & (so == 1)
    assert agree.sum() > (1000 if dynamics != "kbm" else 250)
    if dynamics == "dbm" and not slip_io:
        agree &= np.abs(uo[:, 1]) < 1.2                               # tan(beta) well conditioned on the way back
    if dynamics == "kbm":
        agree &= np.abs(uo[:, 1] * 2.9 / w["u_ref"][:, 0]) < 3.0      # atan2(omega L, v) likewise
    n = _judge(ug[agree], uo[agree], so[agree], t[agree], rows[:, agree], un[agree], R, dtype,
               frac_close=0.95 if dtype == torch.float64 else 0.9)
    assert n > (700 if dynamics != "kbm" else 200)


def test_where_the_flag_is_refused():
    import ctypes as C
    from sccav_cbf_amd import _lib
    lib = _lib.load()
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    p.flags = _lib.FLAG_INFEASIBLE_MIN_VIOLATION
    one = torch.zeros((64,), dtype=torch.float64, device=DEV)
    ptr = C.c_void_p(one.data_ptr())
    assert lib.sccbf_solve_f64(C.byref(p), 4, 2, ptr, ptr, ptr, ptr, ptr, ptr, None, None) == _lib.ERR_ARG   # no rows_out
    assert lib.sccbf_solve_step_f64(C.byref(p), 0.5, 0.1, 4, 2, ptr, ptr, ptr, ptr, ptr, ptr, None, None) == _lib.ERR_ARG
    assert lib.sccbf_solve_sa_f64(C.byref(p), 0.1, 4, 2, ptr, ptr, ptr, ptr, ptr, ptr, ptr, None, None) == _lib.ERR_ARG
    assert lib.sccbf_solve_batches_f64(C.byref(p), 1, 2, 4, ptr, None) == _lib.ERR_ARG
