"""Closed-loop replays built on the CPU oracle (TEST INFRASTRUCTURE ONLY).

replay_stanley_cone() restates the as-shipped loop of the reference driver
test_scripts/stanley_controller_ellipse.py:581-750 (CBF_TYPE = 4, line 620):
Stanley + P speed control as nominal input, one CollisionCone2D row around the
elliptical obstacle, DBM_CBF_2DS(alpha=1) with R = diag(.5,.5), vehicle step
update_com.  Its slip-angle history is what the reference saved to
test_scripts/beta_vs_time.mat (lines 748, 1066-1069).
"""
import os

import numpy as np

from . import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

# test_scripts/stanley_controller_ellipse.py:52-58
K_STANLEY = 0.5
KP = 1.0
DT = 0.1
WHEELBASE = 2.9
LR = WHEELBASE / 2
LF = WHEELBASE - LR
MAX_STEER = np.radians(30.0)


def load_course():
    z = np.load(os.path.join(GOLDEN, "sce_course.npz"))
    return z["cx"], z["cy"], z["cyaw"]


def scenario_constants(cx, cy):
    """Obstacle and cone radius of stanley_controller_ellipse.py:608-612, 645."""
    last_idx = len(cx) - 1
    obs_idx = int(last_idx * 0.75)
    a, b = 20.0, 10.0
    return dict(last_idx=last_idx, o_cx=cx[obs_idx], o_cy=cy[obs_idx], a=a, b=b,
                a_cone=np.hypot(a, b) / 2, target_speed=30.0 / 3.6, max_time=30.0)


def replay_stanley_cone(solve_tick=None, row_kind="cone"):
    """Run the loop; returns (t, beta_deg, log).  solve_tick(s, u_ref, slot, kind) -> (a, delta)
    lets a test drive the same loop through another implementation; default = oracle.
    row_kind "cone" is the as-shipped scenario, "ellipse" constrains with the true
    ellipse row (a=20, b=10) instead."""
    cx, cy, cyaw = load_course()
    k = scenario_constants(cx, cy)
    params = orc.make_params(alpha=1.0, lr=LR, lf=LF, R=(0.5, 0, 0, 0.5))

    def oracle_tick(s, u_ref, slot, kind):
        obs = np.zeros((1, 2, 1, 4))
        obs[0, :, 0, :] = slot.reshape(2, 4)
        u, st, _ = orc.solve_batch(params, s[None, :], np.asarray(u_ref)[None, :], obs,
                                   np.array([kind], dtype=np.uint8))
        return u[0, 0], u[0, 1]

    if solve_tick is None:
        solve_tick = oracle_tick

    s = np.array([0.0, 5.0, np.radians(20.0), 10.0])       # line 595
    _, target_idx = orc.stanley(s, cx, cy, cyaw, 0, WHEELBASE, K_STANLEY)
    time = 0.0
    t_arr, beta = [0.0], [0.0]
    log = []
    while k["max_time"] >= time and k["last_idx"] > target_idx:
        a_ref = KP * (k["target_speed"] - s[3])
        d_ref, target_idx = orc.stanley(s, cx, cy, cyaw, target_idx, WHEELBASE, K_STANLEY)
        slot = np.zeros(8)
        if row_kind == "cone":
            # CollisionCone2D(a_cone, s, s_obs=(o_cx, o_cy, 0, 0)), default buffer 1.5
            slot[:6] = [k["o_cx"], k["o_cy"], 0.0, 0.0, k["a_cone"] + 1.5, 0.0]
            kind = orc.OBS_CONE
        else:
            slot[:7] = [k["o_cx"], k["o_cy"], k["a"], k["b"], 0.0, 0.0, 0.0]
            kind = orc.OBS_ELLIPSE
        a_cbf, d_cbf = solve_tick(s.copy(), (a_ref, d_ref), slot, kind)
        log.append((s.copy(), a_ref, d_ref, a_cbf, d_cbf))
        b = orc.vehicle_step(s, a_cbf, d_cbf, LR, LF, MAX_STEER, DT)
        beta.append(b)
        time += DT
        t_arr.append(time)
    return np.array(t_arr), np.degrees(np.array(beta)), log
