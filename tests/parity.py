"""Helpers shared by the parity tests: compare a filter output with the fp64 oracle."""
import numpy as np


def to_internal(u, params):
    """(a, delta) -> QP coordinates (a, beta) for DBM; (v, delta) -> (v, omega) needs v_ref."""
    ui = np.array(u, dtype=np.float64, copy=True)
    ui[:, 1] = np.arctan2(params.lr * np.tan(ui[:, 1]), params.lf + params.lr)
    return ui


def qp_metrics(u_int, u_nom_int, rows, R):
    """cost and worst relative row violation of u (A,2) against rows (M,A,4), all fp64."""
    d = u_int - u_nom_int
    cost = R[0, 0] * d[:, 0] ** 2 + (R[0, 1] + R[1, 0]) * d[:, 0] * d[:, 1] + R[1, 1] * d[:, 1] ** 2
    lhs = rows[:, :, 0] * u_int[None, :, 0] + rows[:, :, 1] * u_int[None, :, 1]
    scale = (np.hypot(rows[:, :, 0], rows[:, :, 1]) * np.hypot(u_int[:, 0], u_int[:, 1])[None, :]
             + np.abs(rows[:, :, 2]) + 1e-300)
    viol = np.max((rows[:, :, 2] - lhs) / scale, axis=0) if rows.shape[0] else np.zeros(len(cost))
    return cost, viol


def active_det(u_int, rows, tol=1e-6):
    """Conditioning of the active set at u: |sin| of the angle between the two tightest
    rows when two are active, 1 otherwise."""
    lhs = rows[:, :, 0] * u_int[None, :, 0] + rows[:, :, 1] * u_int[None, :, 1]
    nrm = np.hypot(rows[:, :, 0], rows[:, :, 1]) + 1e-300
    slack = (lhs - rows[:, :, 2]) / nrm
    A = u_int.shape[0]
    out = np.ones(A)
    if rows.shape[0] < 2:
        return out
    order = np.argsort(slack, axis=0)
    i0, i1 = order[0], order[1]
    ar = np.arange(A)
    both = (slack[i0, ar] < tol) & (slack[i1, ar] < tol)
    a0 = rows[i0, ar, :2] / nrm[i0, ar, None]
    a1 = rows[i1, ar, :2] / nrm[i1, ar, None]
    det = np.abs(a0[:, 0] * a1[:, 1] - a0[:, 1] * a1[:, 0])
    out[both] = det[both]
    return out
