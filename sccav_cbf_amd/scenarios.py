"""Seeded synthetic workloads in the ABI layout (numpy, float64; cast as needed).

Shapes follow the reference's scenarios:
  radial_seekers   test_scripts/radial_dynamic_obstacles.py:49-62, 151-168, 222-225
  monte_carlo      ego / obstacle distributions around the driver-script scale
                   (test_scripts/stanley_controller_ellipse.py:52-58, 584-612)
  lanes_and_cones  the two straight lane boundaries and vehicle cones of
                   test_scripts/carla_scripts/multi_obstacle_CBF_local_with_lanes.py:276-297, 913-927
Every generator returns a dict: state (A,4), u_ref (A,2), obs (M,2,A,4), obs_type (M,) uint8.
"""
import numpy as np

OBS_ELLIPSE, OBS_CONE, OBS_LANE, OBS_RADIAL, OBS_EMPTY = 0, 1, 2, 3, 255
OBS_ELLIPSE_PREP, OBS_CONE_PREP = 4, 5
OBS_LANE6 = 8

MAX_STEER = np.radians(30.0)
TARGET_SPEED = 30.0 / 3.6
KP = 1.0


def _pack(slots):
    """(M, A, 8) -> (M, 2, A, 4)"""
    M, A, _ = slots.shape
    return np.ascontiguousarray(slots.reshape(M, A, 2, 4).transpose(0, 2, 1, 3))


def prepared(w):
    """The same batch with every ellipse and cone slot in its prepared form (kinds 4 and 5 of
    include/sccbf.h): what does not depend on the agent is evaluated once, here in float64,
        ellipse  cx, cy, a, b, theta_o, vx, vy, -        ->  cx, cy, 1/a^2, 1/b^2, cos, sin, vx, vy
        cone     cx, cy, theta_o, v_o, a, beta, -, -     ->  cx, cy, v_o cos(theta_o + beta), v_o sin(..), a, beta, -, -
    obs (M,2,A,4) and obs_type (M,) or (M,A); other slots are kept."""
    obs = np.array(w["obs"], dtype=np.float64)
    ty = np.array(w["obs_type"], dtype=np.uint8)
    M, _, A, _ = obs.shape
    sl = np.concatenate([obs[:, 0], obs[:, 1]], axis=2)            # (M, A, 8)
    kind = np.broadcast_to(ty.reshape(M, -1), (M, A))
    out = sl.copy()
    e, c = kind == OBS_ELLIPSE, kind == OBS_CONE
    with np.errstate(divide="ignore", invalid="ignore"):
        out[..., 2] = np.where(e, 1.0 / (sl[..., 2] * sl[..., 2]), out[..., 2])
        out[..., 3] = np.where(e, 1.0 / (sl[..., 3] * sl[..., 3]), out[..., 3])
    out[..., 4] = np.where(e, np.cos(sl[..., 4]), out[..., 4])
    out[..., 5] = np.where(e, np.sin(sl[..., 4]), out[..., 5])
    out[..., 6] = np.where(e, sl[..., 5], out[..., 6])
    out[..., 7] = np.where(e, sl[..., 6], out[..., 7])
    ang = sl[..., 2] + sl[..., 5]
    out[..., 2] = np.where(c, sl[..., 3] * np.cos(ang), out[..., 2])
    out[..., 3] = np.where(c, sl[..., 3] * np.sin(ang), out[..., 3])
    ty2 = ty.copy()
    ty2[ty == OBS_ELLIPSE] = OBS_ELLIPSE_PREP
    ty2[ty == OBS_CONE] = OBS_CONE_PREP
    r = dict(w)
    r["obs"] = _pack(out)
    r["obs_type"] = ty2
    return r


def spline_course():
    """Course knots of stanley_controller_ellipse.py:584-585 sampled densely with a plain
    polyline (used only to give the nominal steering of synthetic agents a target)."""
    ax = np.array([0.0, 100.0, 100.0, 50.0, 60.0])
    ay = np.array([0.0, 0.0, -30.0, -20.0, 0.0])
    seg = np.hypot(np.diff(ax), np.diff(ay))
    s = np.concatenate([[0.0], np.cumsum(seg)])
    ss = np.arange(0.0, s[-1], 0.5)
    cx, cy = np.interp(ss, s, ax), np.interp(ss, s, ay)
    cyaw = np.arctan2(np.gradient(cy), np.gradient(cx))
    return cx, cy, cyaw


def stanley_nominal(state, cx, cy, cyaw, axle=2.9, k=0.5, chunk=8192):
    """Vectorised Stanley law (stanley_controller_ellipse.py:146-212) without index memory,
    steering clipped to the actuator range."""
    A = state.shape[0]
    delta = np.empty(A)
    for lo in range(0, A, chunk):
        s = state[lo:lo + chunk]
        fx = s[:, 0] + axle * np.cos(s[:, 2])
        fy = s[:, 1] + axle * np.sin(s[:, 2])
        d2 = (fx[:, None] - cx[None, :]) ** 2 + (fy[:, None] - cy[None, :]) ** 2
        idx = np.argmin(d2, axis=1)
        ex, ey = -np.cos(s[:, 2] + np.pi / 2), -np.sin(s[:, 2] + np.pi / 2)
        err = (fx - cx[idx]) * ex + (fy - cy[idx]) * ey
        th_e = (cyaw[idx] - s[:, 2] + np.pi) % (2 * np.pi) - np.pi
        delta[lo:lo + chunk] = th_e + np.arctan2(k * err, s[:, 3])
    return np.clip(delta, -MAX_STEER, MAX_STEER)


def _ego(rng, A):
    state = np.empty((A, 4))
    state[:, 0] = rng.uniform(-50, 50, A)
    state[:, 1] = rng.uniform(-50, 50, A)
    state[:, 2] = rng.uniform(-np.pi, np.pi, A)
    state[:, 3] = rng.uniform(0, 15, A)
    return state


def _centres(rng, state, n):
    A = state.shape[0]
    rg = rng.uniform(15, 60, (n, A))
    br = rng.uniform(0, 2 * np.pi, (n, A))
    return state[None, :, 0] + rg * np.cos(br), state[None, :, 1] + rg * np.sin(br)


def _ellipse_slots(rng, state, n):
    A = state.shape[0]
    cx, cy = _centres(rng, state, n)
    sl = np.zeros((n, A, 8))
    sl[..., 0], sl[..., 1] = cx, cy
    sl[..., 2] = rng.uniform(2, 20, (n, A))
    sl[..., 3] = rng.uniform(1, 10, (n, A))
    sl[..., 4] = rng.uniform(-np.pi, np.pi, (n, A))
    sp = rng.uniform(0, 5, (n, A))
    sl[..., 5] = sp * np.cos(sl[..., 4])
    sl[..., 6] = sp * np.sin(sl[..., 4])
    return sl


def _cone_slots(rng, state, n):
    A = state.shape[0]
    cx, cy = _centres(rng, state, n)
    sl = np.zeros((n, A, 8))
    sl[..., 0], sl[..., 1] = cx, cy
    sl[..., 2] = rng.uniform(-np.pi, np.pi, (n, A))     # obstacle heading
    sl[..., 3] = rng.uniform(0, 5, (n, A))              # obstacle speed
    sl[..., 4] = rng.uniform(1.0, 5.0, (n, A)) + 1.5    # radius + default buffer 1.5
    return sl


def monte_carlo(A, n_ellipse, n_cone, seed, nominal="stanley"):
    """Monte-Carlo batch: n_ellipse ellipse slots followed by n_cone collision-cone slots."""
    rng = np.random.default_rng(seed)
    state = _ego(rng, A)
    parts, types = [], []
    if n_ellipse:
        parts.append(_ellipse_slots(rng, state, n_ellipse))
        types += [OBS_ELLIPSE] * n_ellipse
    if n_cone:
        parts.append(_cone_slots(rng, state, n_cone))
        types += [OBS_CONE] * n_cone
    slots = np.concatenate(parts, axis=0)
    u_ref = np.empty((A, 2))
    u_ref[:, 0] = KP * (TARGET_SPEED - state[:, 3])
    if nominal == "stanley":
        u_ref[:, 1] = stanley_nominal(state, *spline_course())
    else:
        u_ref[:, 1] = rng.uniform(-MAX_STEER, MAX_STEER, A)
    return dict(state=state, u_ref=u_ref, obs=_pack(slots), obs_type=np.array(types, dtype=np.uint8))


def config3(A=4096, seed=1234 + 3):
    """4096 agents x 8 ellipses, Stanley nominal."""
    return monte_carlo(A, 8, 0, seed)


def config4(A=65536, seed=1234 + 4, nominal="stanley"):
    """65536 agents x 16 mixed rows (8 ellipse + 8 cone)."""
    return monte_carlo(A, 8, 8, seed, nominal=nominal)


def radial_seekers(M=32, seed=1234 + 2, A=1):
    """Ego at rest at the origin, M circular seekers on the annulus, closing at
    max(0.2 dist, 3) m/s; kv = 1 (radial_dynamic_obstacles.py:444, 462-463)."""
    rng = np.random.default_rng(seed)
    state = np.zeros((A, 4))
    r = rng.uniform(1.5, 2.0, (M, A))
    rg = rng.uniform(10.0, 20.0, (M, A))
    br = rng.uniform(0, 2 * np.pi, (M, A))
    sl = np.zeros((M, A, 8))
    sl[..., 0], sl[..., 1] = rg * np.cos(br), rg * np.sin(br)
    sl[..., 2] = r
    sl[..., 3] = r
    sp = np.maximum(0.2 * rg, 3.0)
    sl[..., 4] = -sp * np.cos(br)
    sl[..., 5] = -sp * np.sin(br)
    sl[..., 6] = 1.0
    return dict(state=state, u_ref=np.zeros((A, 2)), obs=_pack(sl),
                obs_type=np.full(M, OBS_RADIAL, dtype=np.uint8))


def lanes_and_cones(A=16384, n_cone=6, seed=1234 + 5, lane_y=(19.4, 10.4), buffer=1.5, curved=False):
    """Two straight lane boundaries (degree-1 PolyLane, default buffer 1.5) plus n_cone
    vehicle cones; ego between the lanes heading roughly along +x.  curved: the boundaries are gently curved
    polynomials of degree 5 and 6 instead (kind 8: c0 ... c6, buffer), a few decimetres off the straight ones over
    the stretch the agents are on."""
    rng = np.random.default_rng(seed)
    state = np.empty((A, 4))
    state[:, 0] = rng.uniform(-95, 0, A)
    state[:, 1] = rng.uniform(lane_y[1] + 1.6, lane_y[0] - 1.6, A)
    state[:, 2] = rng.uniform(-0.3, 0.3, A)
    state[:, 3] = rng.uniform(1, 15, A)
    lanes = np.zeros((2, A, 8))
    lanes[0, :, 0] = lane_y[0]
    lanes[1, :, 0] = lane_y[1]
    lanes[..., 4] = buffer
    cones = _cone_slots(rng, state, n_cone)
    u_ref = np.empty((A, 2))
    u_ref[:, 0] = KP * (TARGET_SPEED - state[:, 3])
    u_ref[:, 1] = rng.uniform(-0.2, 0.2, A)
    lane_kind = OBS_LANE
    if curved:
        lane_kind = OBS_LANE6
        lanes[..., 4] = 0.0
        lanes[..., 7] = buffer
        lanes[0, :, 1:6] = [2e-3, -4e-5, 3e-7, 2e-9, -1e-11]              # degree 5
        lanes[1, :, 1:7] = [-1e-3, 3e-5, 2e-7, -3e-9, 2e-11, 1e-13]       # degree 6
    types = np.array([lane_kind, lane_kind] + [OBS_CONE] * n_cone, dtype=np.uint8)
    return dict(state=state, u_ref=u_ref, obs=_pack(np.concatenate([lanes, cones], axis=0)),
                obs_type=types)
