"""ctypes front-end of the CPU oracle (oracle/cbf_oracle.c).

TEST INFRASTRUCTURE ONLY: the parity checker and the timed CPU baseline.  Only
tests/, the smoke check and bench.py's cpu_baseline leg import this module; the
product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

OBS_ELLIPSE, OBS_CONE, OBS_LANE, OBS_RADIAL, OBS_EMPTY = 0, 1, 2, 3, 255
OBS_LANE6 = 8   # a lane polynomial of degree up to 6: c0 ... c6, buffer
OBS_ELLIPSE_PREP, OBS_CONE_PREP = 4, 5
OBS_ELLIPSE_BOX, OBS_CONE_BOX = 6, 7   # the slot is the bounding box itself: x, y, extent x, y | yaw, speed, buffer
DYN_DBM, DYN_KBM, DYN_DUM, DYN_SADBM = 0, 1, 2, 3
FLAG_OBS_SHARED, FLAG_TYPE_PER_AGENT, FLAG_SLIP_IO = 1, 2, 4
FLAG_CONE_BETA_STATE = 16


class Params(C.Structure):
    """Field-for-field the sccbf_params of include/sccbf.h."""
    _fields_ = [("alpha", C.c_double), ("lr", C.c_double), ("lf", C.c_double), ("L", C.c_double),
                ("R", C.c_double * 4), ("zero_tol", C.c_double), ("dynamics", C.c_int32),
                ("lanes_per_qp", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_int32)]


class Partials(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("h", "hx", "hy", "hth", "hv", "ht")]


def make_params(alpha=1.0, lr=1.45, lf=1.45, L=2.9, R=(1.0, 0.0, 0.0, 1.0), zero_tol=1e-3,
                dynamics=DYN_DBM, flags=0, lanes_per_qp=0):
    p = Params()
    p.alpha, p.lr, p.lf, p.L = alpha, lr, lf, L
    R = np.asarray(R, dtype=np.float64).reshape(-1)
    for i in range(4):
        p.R[i] = float(R[i])
    p.zero_tol, p.dynamics, p.flags, p.lanes_per_qp = zero_tol, dynamics, flags, lanes_per_qp
    return p


def build(force=False):
    """Compile liboracle.so with the Makefile next to this file."""
    src = os.path.join(_HERE, "cbf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        L.orc_solve_batch.restype = C.c_int
        L.orc_solve_batch.argtypes = [C.POINTER(Params), C.c_int64, C.c_int32, dp, dp, dp,
                                      C.POINTER(C.c_uint8), dp, C.POINTER(C.c_int32), dp, C.c_int]
        L.orc_solve_sa_batch.restype = C.c_int
        L.orc_solve_sa_batch.argtypes = [C.POINTER(Params), C.c_double, C.c_int64, C.c_int32, dp, dp, dp, dp,
                                         C.POINTER(C.c_uint8), dp, C.POINTER(C.c_int32), dp, C.c_int]
        L.orc_row_sa.restype = None
        L.orc_row_sa.argtypes = [C.POINTER(Params), dp, C.c_double, C.POINTER(Partials), C.c_double, dp]
        L.orc_qp_batch.restype = C.c_int
        L.orc_qp_batch.argtypes = [C.POINTER(Params), C.c_int64, C.c_int32, dp, dp, dp,
                                   C.POINTER(C.c_int32), C.c_int]
        L.orc_barrier.restype = C.c_int
        L.orc_barrier.argtypes = [C.c_int, dp, dp, C.c_double, C.POINTER(Partials)]
        L.orc_row.restype = None
        L.orc_row.argtypes = [C.POINTER(Params), dp, C.POINTER(Partials), dp]
        L.orc_lane_closest_x.restype = C.c_double
        L.orc_lane_closest_x.argtypes = [dp, C.c_double, C.c_double]
        L.orc_lane_closest_x_n.restype = C.c_double
        L.orc_lane_closest_x_n.argtypes = [dp, C.c_int, C.c_double, C.c_double]
        L.orc_vehicle_step.restype = C.c_double
        L.orc_vehicle_step.argtypes = [dp] + [C.c_double] * 6
        L.orc_stanley.restype = C.c_double
        L.orc_stanley.argtypes = [dp, C.c_int, dp, dp, dp, C.c_int, C.c_double, C.c_double,
                                  C.c_double, C.POINTER(C.c_int)]
        L.orc_nominal_batch.restype = None
        L.orc_nominal_batch.argtypes = [C.c_int64, C.c_int32, dp, dp, C.POINTER(C.c_int32), dp] + \
            [C.c_double] * 8 + [dp]
        L.orc_vehicle_step_batch.restype = None
        L.orc_vehicle_step_batch.argtypes = [C.c_int64, dp, dp] + [C.c_double] * 4 + [dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def solve_batch(params, state, u_ref, obs, obs_type, want_rows=False, incremental=False):
    """One tick for A agents.  state (A,4), u_ref (A,2), obs (M,2,A,4) or (M,2,4) when
    FLAG_OBS_SHARED, obs_type (M,) or (M,A) uint8.  Returns u_out (A,2), status (A,),
    rows (M,A,4) or None."""
    state, u_ref, obs = _f64(state), _f64(u_ref), _f64(obs)
    obs_type = np.ascontiguousarray(obs_type, dtype=np.uint8)
    A, M = state.shape[0], obs.shape[0]
    assert state.shape == (A, 4) and u_ref.shape == (A, 2)
    if params.flags & FLAG_OBS_SHARED:
        assert obs.shape == (M, 2, 4)
    else:
        assert obs.shape == (M, 2, A, 4)
    assert obs_type.shape == ((M, A) if params.flags & FLAG_TYPE_PER_AGENT else (M,))
    u_out = np.empty((A, 2))
    status = np.empty(A, dtype=np.int32)
    rows = np.empty((M, A, 4)) if want_rows else None
    rc = lib().orc_solve_batch(C.byref(params), A, M, _dp(state), _dp(u_ref), _dp(obs),
                               obs_type.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(u_out),
                               status.ctypes.data_as(C.POINTER(C.c_int32)),
                               _dp(rows) if want_rows else None, int(incremental))
    if rc != 0:
        raise ValueError("orc_solve_batch failed: %d" % rc)
    return u_out, status, rows


def solve_sa_batch(params, dt, state, slip, u_ref, obs, obs_type, want_rows=False, incremental=False):
    """One tick of the steer-rate model (SADBM_CBF_2DS.solve_cbf, cbf/cbf.py:349-437) for A agents.
    slip (A,2) float64 = beta, beta_ref_last is updated in place.  Returns u_out (A,2) = a*, delta*,
    status (A,), rows (M,A,4) or None."""
    state, u_ref, obs = _f64(state), _f64(u_ref), _f64(obs)
    obs_type = np.ascontiguousarray(obs_type, dtype=np.uint8)
    A, M = state.shape[0], obs.shape[0]
    assert slip.dtype == np.float64 and slip.shape == (A, 2) and slip.flags.c_contiguous
    assert obs.shape == ((M, 2, 4) if params.flags & FLAG_OBS_SHARED else (M, 2, A, 4))
    assert obs_type.shape == ((M, A) if params.flags & FLAG_TYPE_PER_AGENT else (M,))
    u_out = np.empty((A, 2))
    status = np.empty(A, dtype=np.int32)
    rows = np.empty((M, A, 4)) if want_rows else None
    rc = lib().orc_solve_sa_batch(C.byref(params), float(dt), A, M, _dp(state), _dp(slip), _dp(u_ref), _dp(obs),
                                  obs_type.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(u_out),
                                  status.ctypes.data_as(C.POINTER(C.c_int32)),
                                  _dp(rows) if want_rows else None, int(incremental))
    if rc != 0:
        raise ValueError("orc_solve_sa_batch failed: %d" % rc)
    return u_out, status, rows


def row_sa(params, s, beta, partials, h_beta):
    d = Partials(**partials)
    s = _f64(s)
    out = np.empty(3)
    lib().orc_row_sa(C.byref(params), _dp(s), float(beta), C.byref(d), float(h_beta), _dp(out))
    return out


def qp_batch(params, rows, u_nom, incremental=False):
    """QP stage alone.  rows (M,A,4) [a0,a1,b,-], u_nom (A,2) -> u (A,2), status (A,)."""
    rows, u_nom = _f64(rows), _f64(u_nom)
    M, A = rows.shape[0], rows.shape[1]
    assert rows.shape == (M, A, 4) and u_nom.shape == (A, 2)
    u_out = np.empty((A, 2))
    status = np.empty(A, dtype=np.int32)
    rc = lib().orc_qp_batch(C.byref(params), A, M, _dp(rows), _dp(u_nom), _dp(u_out),
                            status.ctypes.data_as(C.POINTER(C.c_int32)), int(incremental))
    if rc != 0:
        raise ValueError("orc_qp_batch failed: %d" % rc)
    return u_out, status


def qp_min_violation_batch(params, rows, u_nom):
    """QP stage with the least-violation answer where the rows admit no input: rows (M,A,4), u_nom (A,2) ->
    u (A,2), status (A,) (1 stays 1), t (A,) = the least worst-case violation of those agents (distance to the
    row's half-plane in the metric of the cost), 0 for the others.  Enumeration over row triples: M <= 64."""
    rows, u_nom = _f64(rows), _f64(u_nom)
    M, A = rows.shape[0], rows.shape[1]
    assert rows.shape == (M, A, 4) and u_nom.shape == (A, 2)
    u_out, t = np.empty((A, 2)), np.empty(A)
    status = np.empty(A, dtype=np.int32)
    L = lib()
    L.orc_qp_min_violation_batch.restype = C.c_int
    L.orc_qp_min_violation_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_double)]
    rc = L.orc_qp_min_violation_batch(C.byref(params), A, M, _dp(rows), _dp(u_nom), _dp(u_out),
                                      status.ctypes.data_as(C.POINTER(C.c_int32)), _dp(t))
    if rc != 0:
        raise ValueError("orc_qp_min_violation_batch failed: %d" % rc)
    return u_out, status, t


def barrier(kind, s, o, zero_tol=1e-3):
    """h and partials of one obstacle slot: returns dict h,hx,hy,hth,hv,ht."""
    s, o8 = _f64(s), np.zeros(8)
    o = _f64(o).reshape(-1)
    o8[:o.size] = o
    d = Partials()
    lib().orc_barrier(int(kind), _dp(s), _dp(o8), zero_tol, C.byref(d))
    return {n: getattr(d, n) for n, _ in Partials._fields_}


def row(params, s, partials):
    d = Partials(**partials)
    s = _f64(s)
    out = np.empty(3)
    lib().orc_row(C.byref(params), _dp(s), C.byref(d), _dp(out))
    return out


def lane_closest_x(coeffs, px, py):
    """Abscissa of the curve point nearest to (px, py); up to 4 coefficients through the cubic form, up to 7 through
    the general one."""
    cc = _f64(coeffs).reshape(-1)
    if cc.size > 4:
        c = np.zeros(7)
        c[:cc.size] = cc
        return lib().orc_lane_closest_x_n(_dp(c), 7, float(px), float(py))
    c = np.zeros(4)
    c[:cc.size] = cc
    return lib().orc_lane_closest_x(_dp(c), float(px), float(py))


def vehicle_step(s, acc, delta, lr, lf, max_steer, dt):
    """In-place bicycle step on s (4,) float64; returns beta."""
    assert s.dtype == np.float64 and s.flags.c_contiguous
    return lib().orc_vehicle_step(_dp(s), acc, delta, lr, lf, max_steer, dt)


def stanley(s, cx, cy, cyaw, last_idx, axle, k, ks=0.0):
    s, cx, cy, cyaw = _f64(s), _f64(cx), _f64(cy), _f64(cyaw)
    idx = C.c_int(0)
    d = lib().orc_stanley(_dp(s), cx.size, _dp(cx), _dp(cy), _dp(cyaw), int(last_idx), axle, k, ks,
                          C.byref(idx))
    return d, idx.value


def nominal_batch(state, traj, target_idx, pid_state=None, axle=2.9, k=0.5, ks=0.0, kp=1.0, ki=0.0,
                  kd=0.0, dt=0.1, target_speed=30.0 / 3.6):
    """Stanley + PID1 for a batch on one course.  state (A,4), traj (N,4) = x, y, yaw, v_des,
    target_idx (A,) int32 and pid_state (A,2) are updated in place.  Returns u_ref (A,2)."""
    state, traj = _f64(state), _f64(traj)
    A, N = state.shape[0], traj.shape[0]
    assert target_idx.dtype == np.int32 and target_idx.shape == (A,) and target_idx.flags.c_contiguous
    if pid_state is not None:
        assert pid_state.dtype == np.float64 and pid_state.shape == (A, 2) and pid_state.flags.c_contiguous
    u = np.empty((A, 2))
    lib().orc_nominal_batch(A, N, _dp(state), _dp(traj), target_idx.ctypes.data_as(C.POINTER(C.c_int32)),
                            _dp(pid_state) if pid_state is not None else None, axle, k, ks, kp, ki, kd, dt,
                            target_speed, _dp(u))
    return u


def vehicle_step_batch(state, u, lr, lf, max_steer, dt):
    """In-place bicycle step on state (A,4) float64 with u (A,2) = a, delta; returns beta (A,)."""
    assert state.dtype == np.float64 and state.flags.c_contiguous
    u = _f64(u)
    beta = np.empty(state.shape[0])
    lib().orc_vehicle_step_batch(state.shape[0], _dp(state), _dp(u), lr, lf, max_steer, dt, _dp(beta))
    return beta


def normalize_angle(a):
    """cbf/utils.py:93-106: add / subtract 2 pi until the angle lies in [-pi, pi]."""
    a = np.array(a, dtype=np.float64)
    a -= 2 * np.pi * np.rint(a / (2 * np.pi))
    a = np.where(a > np.pi, a - 2 * np.pi, a)
    return np.where(a < -np.pi, a + 2 * np.pi, a)


def vehicle_step_kin_batch(state, u, mode, L, max_steer, dt):
    """In place on state (A,4) float64: State.update (mode "acc", u = (a, delta)) and
    State.update_by_vel ("vel", u = (v, delta)) of reference
    test_scripts/stanley_controller_ellipse.py:86-118, and the Euler step of the unicycle that
    DUM_CBF_2DS filters ("unicycle", u = (a, omega), cbf/cbf.py:237-245)."""
    assert state.dtype == np.float64
    u = _f64(u)
    x, y, yaw, v = (state[:, i].copy() for i in range(4))
    state[:, 0] = x + v * np.cos(yaw) * dt
    state[:, 1] = y + v * np.sin(yaw) * dt
    if mode == "unicycle":
        state[:, 2] = yaw + u[:, 1] * dt
        state[:, 3] = v + u[:, 0] * dt
        return
    delta = np.clip(u[:, 1], -max_steer, max_steer)
    state[:, 2] = normalize_angle(yaw + v / L * np.tan(delta) * dt)
    state[:, 3] = u[:, 0] if mode == "vel" else v + u[:, 0] * dt


def seeker_step(obs, obs_type, ego_xy, dt, k=0.2, v_min=3.0):
    """ObstacleSpawner.update_seekers, reference test_scripts/radial_dynamic_obstacles.py:210-234,
    for every ellipse and radial slot of obs (M,2,A,4) float64, in place, in the reference's order:
    yaw = arctan2(ego - centre) (217; the ellipse's theta becomes yaw, obstacles.py:288-292),
    speed max(k * distance, v_min) (222-225), velocity = speed (cos yaw, sin yaw)
    (obstacles.py:272-277), centre += velocity dt (233-234).  Cones move with their constant
    velocity along theta_o + beta, lanes stay.  ego_xy (A,2) is the ego position after its step
    (radial_dynamic_obstacles.py:481-487)."""
    assert obs.dtype == np.float64
    ego_xy = _f64(ego_xy)
    for m, kind in enumerate(np.asarray(obs_type).tolist()):
        if kind in (OBS_ELLIPSE, OBS_RADIAL):
            cx, cy = obs[m, 0, :, 0], obs[m, 0, :, 1]
            yaw = np.arctan2(ego_xy[:, 1] - cy, ego_xy[:, 0] - cx)
            v = np.maximum(k * np.hypot(ego_xy[:, 0] - cx, ego_xy[:, 1] - cy), v_min)
            vx, vy = v * np.cos(yaw), v * np.sin(yaw)
            if kind == OBS_ELLIPSE:
                obs[m, 1, :, 0], obs[m, 1, :, 1], obs[m, 1, :, 2] = yaw, vx, vy
            else:
                obs[m, 1, :, 0], obs[m, 1, :, 1] = vx, vy
            cx += vx * dt
            cy += vy * dt
        elif kind == OBS_ELLIPSE_PREP:   # the same ellipse, slot = cx, cy, 1/a^2, 1/b^2 | cos, sin, vx, vy
            cx, cy = obs[m, 0, :, 0], obs[m, 0, :, 1]
            yaw = np.arctan2(ego_xy[:, 1] - cy, ego_xy[:, 0] - cx)
            v = np.maximum(k * np.hypot(ego_xy[:, 0] - cx, ego_xy[:, 1] - cy), v_min)
            vx, vy = v * np.cos(yaw), v * np.sin(yaw)
            obs[m, 1, :, 0], obs[m, 1, :, 1], obs[m, 1, :, 2], obs[m, 1, :, 3] = np.cos(yaw), np.sin(yaw), vx, vy
            cx += vx * dt
            cy += vy * dt
        elif kind == OBS_CONE:
            ang = obs[m, 0, :, 2] + obs[m, 1, :, 1]
            obs[m, 0, :, 0] += obs[m, 0, :, 3] * np.cos(ang) * dt
            obs[m, 0, :, 1] += obs[m, 0, :, 3] * np.sin(ang) * dt
        elif kind == OBS_CONE_PREP:
            obs[m, 0, :, 0] += obs[m, 0, :, 2] * dt
            obs[m, 0, :, 1] += obs[m, 0, :, 3] * dt
        elif kind == OBS_CONE_BOX:       # theta_o = 0: along x at the box's speed (an ellipse box is at rest)
            obs[m, 0, :, 0] += obs[m, 1, :, 1] * dt


def ingest_bbox(bbox, kind, buffer=0.5):
    """Slots (M,2,A,4) from bounding boxes (M,2,A,4) = location x, y, extent x, y | yaw, speed:
    Ellipse2D.from_bounding_box (cbf/obstacles.py:320-331: a = extent.x, b = extent.y, centre,
    theta = yaw, buffer added to both axes by the constructor, 146-162) and
    CollisionCone2D.from_bounding_box (534-543: a = hypot(extent.x, extent.y) + buffer (357),
    s_obs = [x, y, 0, velocity]).  kind (M,) or (M,A)."""
    bbox = _f64(bbox)
    M, A = bbox.shape[0], bbox.shape[2]
    kind = np.broadcast_to(np.asarray(kind, dtype=np.uint8).reshape(M, -1), (M, A))
    obs = np.zeros((M, 2, A, 4))
    x, y, ex, ey, yaw, sp = (bbox[:, 0, :, 0], bbox[:, 0, :, 1], bbox[:, 0, :, 2], bbox[:, 0, :, 3],
                             bbox[:, 1, :, 0], bbox[:, 1, :, 1])
    e, c = kind == OBS_ELLIPSE, kind == OBS_CONE
    ep, cp = kind == OBS_ELLIPSE_PREP, kind == OBS_CONE_PREP   # the same two, per-obstacle constants evaluated
    obs[:, 0, :, 0] = np.where(e | c | ep | cp, x, 0.0)
    obs[:, 0, :, 1] = np.where(e | c | ep | cp, y, 0.0)
    obs[:, 0, :, 2] = np.where(e, ex + buffer, np.where(ep, 1.0 / (ex + buffer) ** 2, np.where(cp, sp, 0.0)))
    obs[:, 0, :, 3] = np.where(e, ey + buffer, np.where(c, sp, np.where(ep, 1.0 / (ey + buffer) ** 2, 0.0)))
    obs[:, 1, :, 0] = np.where(e, yaw, np.where(c | cp, np.hypot(ex, ey) + buffer, np.where(ep, np.cos(yaw), 0.0)))
    obs[:, 1, :, 1] = np.where(ep, np.sin(yaw), 0.0)
    return obs


def lane_fit(px, py, sigma=None, degree=3):
    """PolyLane.fit_polynomial_curve, cbf/obstacles.py:715-773, per agent: the coefficients
    (A, 4), lowest power first, that minimise sum(((p(x) - y) / sigma)^2): the objective of the
    scipy.optimize.curve_fit call at 768-772 for the polynomial model of 765-766.  px, py, sigma
    are (K, A); sigma None = 10 everywhere (737-738); points with sigma <= 0 are left out."""
    px, py = _f64(px), _f64(py)
    K, A = px.shape
    sg = np.full((K, A), 10.0) if sigma is None else _f64(sigma)
    out = np.zeros((A, 4))
    for a in range(A):
        use = sg[:, a] > 0
        x, y, s = px[use, a], py[use, a], sg[use, a]
        xc, hs = 0.5 * (x.min() + x.max()), 0.5 * (x.max() - x.min()) or 1.0
        V = np.vander((x - xc) / hs, degree + 1, increasing=True) / s[:, None]
        d, *_ = np.linalg.lstsq(V, y / s, rcond=None)
        p = np.polynomial.Polynomial(d, domain=[xc - hs, xc + hs], window=[-1, 1]).convert(
            domain=[-1, 1], window=[-1, 1])
        out[a, :p.coef.size] = p.coef
    return out
