#!/usr/bin/env python3
"""Frozen vectors from the build's own fp64 oracle (oracle/cbf_oracle.c), written to tests/golden/oracle_cases.npz.

Not reference data: the reference holds one numeric artifact on this path (beta_vs_time.mat, see
make_reference_fixtures.py).  These vectors pin the ORACLE to itself, so that a later change to it cannot move
the answers every GPU parity test is measured against without a test noticing, and they give the HIP path a
fixed set of inputs that includes the degenerate cases of every obstacle kind:

  barriers  per kind (ellipse, cone, lane, radial, lane of degree up to 6): ego state (4), slot (8) -> h, dh/dx, dh/dy, dh/dtheta,
            dh/dv, dh/dt; random cases plus: the ego at the ellipse's centre; a cone seen from inside its
            radius, from its rim, from its centre, at the ego's own velocity (relative speed zero) and with a
            slip offset; a lane point whose eta (cbf/obstacles.py:633-636) falls below ZERO_TOL; a radial
            obstacle around an ego at rest
  filters   64 agents x 16 mixed slots, 64 agents x 2 lanes + 6 cones and the same with lane polynomials of degree 5
            and 6 (kind 8), for every vehicle model (DBM in
            steering and in slip coordinates, KBM, DUM, SADBM with its slip state) -> rows, input, status
  qp        hand-made row sets (duplicates, parallel and nearly parallel rows, infeasible pair, equality,
            0.u >= 1, empty row, vertex-active and empty triangles, badly scaled rows, 16 tangents to a circle)
            and 256 random sets of 5 rows, a third of them infeasible -> input, status

    python tests/golden/make_oracle_fixtures.py          # rewrites the file
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402
from sccav_cbf_amd import scenarios  # noqa: E402

OUT = os.path.join(HERE, "oracle_cases.npz")
R_HALF = (0.5, 0.0, 0.0, 0.5)


def barrier_cases():
    rng = np.random.default_rng(20261114)
    kinds, states, slots = [], [], []

    def add(kind, s, o):
        o8 = np.zeros(8)
        o8[:len(o)] = o
        kinds.append(kind)
        states.append(np.asarray(s, dtype=np.float64))
        slots.append(o8)

    def ego():
        return [rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-np.pi, np.pi), rng.uniform(0, 15)]

    for _ in range(64):   # ellipse: cx, cy, a, b, theta_o, vx, vy
        s = ego()
        rg, br = rng.uniform(5, 60), rng.uniform(-np.pi, np.pi)
        add(orc.OBS_ELLIPSE, s, [s[0] + rg * np.cos(br), s[1] + rg * np.sin(br), rng.uniform(2, 20), rng.uniform(1, 10),
                                 rng.uniform(-np.pi, np.pi), rng.uniform(-5, 5), rng.uniform(-5, 5)])
    add(orc.OBS_ELLIPSE, [3.0, -2.0, 0.3, 4.0], [3.0, -2.0, 6.0, 2.0, 0.7, 1.0, -1.0])       # at the centre
    add(orc.OBS_ELLIPSE, [0.0, 0.0, 0.0, 0.0], [10.0, 0.0, 4.0, 4.0, 0.0, 0.0, 0.0])         # circle, ego at rest
    for _ in range(64):   # cone: cx, cy, theta_o, v_o, a, beta
        s = ego()
        rg, br = rng.uniform(5, 60), rng.uniform(-np.pi, np.pi)
        add(orc.OBS_CONE, s, [s[0] + rg * np.cos(br), s[1] + rg * np.sin(br), rng.uniform(-np.pi, np.pi), rng.uniform(0, 5),
                              rng.uniform(2, 8), 0.0 if rng.uniform() < 0.7 else rng.uniform(-0.3, 0.3)])
    add(orc.OBS_CONE, [0.0, 0.0, 0.2, 6.0], [2.0, 1.0, 1.0, 3.0, 5.0, 0.0])                   # inside the radius
    add(orc.OBS_CONE, [0.0, 0.0, 0.2, 6.0], [3.0, 4.0, 1.0, 3.0, 5.0, 0.0])                   # on the rim, d = a
    add(orc.OBS_CONE, [1.0, 1.0, -0.4, 6.0], [1.0, 1.0, 2.0, 1.0, 3.0, 0.0])                  # at the centre, d = 0
    add(orc.OBS_CONE, [1.0, 1.0, -0.4, 6.0], [1.0 + 4e-4, 1.0, 2.0, 1.0, 3.0, 0.0])           # d below ZERO_TOL
    add(orc.OBS_CONE, [0.0, 0.0, 0.5, 4.0], [30.0, 5.0, 0.5, 4.0, 3.0, 0.0])                  # the ego's own velocity
    add(orc.OBS_CONE, [0.0, 0.0, 0.5, 4.0], [30.0, 5.0, 0.2, 4.0, 3.0, 0.3])                  # the same through beta
    add(orc.OBS_CONE, [0.0, 0.0, -2.578e-4, 7.0019], [32.76, -7.92, 0.0, 6.99298, 3.27, 0.0])  # 0.009 m/s apart
    for _ in range(48):   # lane: c0..c3, buffer
        s = ego()
        c = [s[1] + rng.uniform(-6, 6), rng.uniform(-0.3, 0.3), rng.uniform(-0.01, 0.01), rng.uniform(-1e-4, 1e-4)]
        c[0] -= c[1] * s[0] + c[2] * s[0] ** 2 + c[3] * s[0] ** 3
        add(orc.OBS_LANE, s, c + [1.5])
    add(orc.OBS_LANE, [5.0, 13.0, 0.1, 8.0], [17.4, 0.0, 0.0, 0.0, 1.5])                      # the shipped straight lane
    # eta = 1 + g' g'' + g'^2 - y g'' below ZERO_TOL (not the centre of curvature: the foot point is well defined):
    # y = 5 + 0.05 x^2 seen from the normal through its point at x = 1, where y g'' = 1 + g' g'' + g'^2
    add(orc.OBS_LANE, [0.485, 10.2, 0.0, 5.0], [5.0, 0.0, 0.05, 0.0, 1.5])
    for _ in range(32):   # radial: cx, cy, a, b, cdot_x, cdot_y, kv
        s = ego()
        rg, br = rng.uniform(5, 30), rng.uniform(-np.pi, np.pi)
        r = rng.uniform(1.5, 2.0)
        add(orc.OBS_RADIAL, s, [s[0] + rg * np.cos(br), s[1] + rg * np.sin(br), r, r, rng.uniform(-4, 4), rng.uniform(-4, 4), 1.0])
    add(orc.OBS_RADIAL, [0.0, 0.0, 0.0, 0.0], [12.0, -3.0, 1.7, 1.7, -3.0, 0.75, 1.0])       # the scenario's ego at rest
    # lanes of degree 4 ... 6 (a generator of their own, so that the cases above stay what they were): c0 ... c6, buffer
    rng = np.random.default_rng(20261115)
    for i in range(36):
        s = ego()
        deg = 4 + i % 3
        c = np.zeros(7)
        c[1:deg + 1] = [rng.uniform(-0.3, 0.3), rng.uniform(-0.01, 0.01), rng.uniform(-1e-4, 1e-4), rng.uniform(-2e-6, 2e-6),
                        rng.uniform(-3e-8, 3e-8), rng.uniform(-4e-10, 4e-10)][:deg]
        c[0] = s[1] + rng.uniform(-6, 6) - np.polynomial.polynomial.polyval(s[0], c)
        add(orc.OBS_LANE6, s, list(c) + [1.5])
    add(orc.OBS_LANE6, [5.0, 13.0, 0.1, 8.0], [17.4, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.5])     # the shipped straight lane in this form
    add(orc.OBS_LANE6, [0.485, 10.2, 0.0, 5.0], [5.0, 0.0, 0.05, 0.0, 0.0, 0.0, 0.0, 1.5])   # eta below ZERO_TOL, as above
    add(orc.OBS_LANE6, [2.0, 3.0, 0.5, 6.0], [1.0, 0.2, -0.03, 0.004, 5e-4, -6e-5, 7e-6, 0.8])   # every coefficient at work
    kinds = np.asarray(kinds, dtype=np.uint8)
    states, slots = np.stack(states), np.stack(slots)
    out = np.stack([[orc.barrier(int(k), s, o)[n] for n in ("h", "hx", "hy", "hth", "hv", "ht")]
                    for k, s, o in zip(kinds, states, slots)])
    return dict(bar_kind=kinds, bar_state=states, bar_slot=slots, bar_out=out)


def filter_cases():
    out = {}
    batches = {"mixed16": scenarios.config4(A=64, nominal="uniform", seed=11),
               "lanes2_cones6": scenarios.lanes_and_cones(A=64, seed=12),
               "lanes6_cones6": scenarios.lanes_and_cones(A=64, seed=13, curved=True)}
    rng = np.random.default_rng(7)
    for name, w in batches.items():
        for key in ("state", "u_ref", "obs", "obs_type"):
            out["%s_%s" % (name, key)] = np.ascontiguousarray(w[key])
        yaw_rate = w["u_ref"].copy()
        yaw_rate[:, 1] *= 3.0
        slip0 = np.stack([rng.uniform(-0.2, 0.2, 64), rng.uniform(-0.2, 0.2, 64)], axis=1)
        out["%s_slip_in" % name] = slip0.copy()
        out["%s_u_ref_dum" % name] = yaw_rate
        for model, p, ur in (("dbm", orc.make_params(R=R_HALF), w["u_ref"]),
                             ("dbm_slip_io", orc.make_params(R=R_HALF, flags=orc.FLAG_SLIP_IO), w["u_ref"]),
                             ("kbm", orc.make_params(R=R_HALF, dynamics=orc.DYN_KBM), w["u_ref"]),
                             ("dum", orc.make_params(R=R_HALF, dynamics=orc.DYN_DUM), yaw_rate)):
            u, st, rows = orc.solve_batch(p, w["state"], ur, w["obs"], w["obs_type"], want_rows=True)
            out["%s_%s_u" % (name, model)], out["%s_%s_status" % (name, model)] = u, st
            out["%s_%s_rows" % (name, model)] = rows
        slip = slip0.copy()
        p = orc.make_params(R=R_HALF, dynamics=orc.DYN_SADBM, flags=orc.FLAG_CONE_BETA_STATE)
        u, st, rows = orc.solve_sa_batch(p, 0.05, w["state"], slip, w["u_ref"], w["obs"], w["obs_type"], want_rows=True)
        out["%s_sadbm_u" % name], out["%s_sadbm_status" % name], out["%s_sadbm_rows" % name] = u, st, rows
        out["%s_sadbm_slip_out" % name] = slip
    return out


def hand_made_rows():
    cases = [
        [(1, 0, 1.0), (1, 0, 1.0), (2, 0, 2.0), (1, 1e-13, 1.0), (0, 1, -3)],     # duplicates / parallel
        [(1, 0, 1.0), (-1, 0, 0.5)],                                               # infeasible pair
        [(1, 0, 1.0), (-1, 0, -1.0)],                                              # equality x = 1
        [(0, 0, 1.0)],                                                             # 0.u >= 1
        [(0, 0, -1.0), (1, 1, 1.0)],                                               # empty row
        [(1, 0, 1.0), (0, 1, 2.0), (-1, -1, -3.5)],                                # triangle, vertex active
        [(1, 0, 1.0), (0, 1, 2.0), (-1, -1, -2.9)],                                # empty triangle
        [(1, 1e-3, 1.0), (1, -1e-3, 1.0)],                                         # nearly parallel
        [(1e-6, 0, 1e-6), (0, 1e6, -1e6)],                                         # badly scaled
        [(np.cos(t), np.sin(t), 1.0) for t in np.linspace(0, np.pi / 2, 16)],     # 16 tangents to a circle
    ]
    M = max(len(c) for c in cases)
    rows = np.zeros((M, len(cases), 4))
    rows[:, :, 2] = -1.0
    for a, c in enumerate(cases):
        for m, (a0, a1, b) in enumerate(c):
            rows[m, a, :3] = (a0, a1, b)
    un = np.zeros((len(cases), 2))
    un[5] = [0.2, 0.1]
    return rows, un


def qp_cases():
    rows, un = hand_made_rows()
    u, st = orc.qp_batch(orc.make_params(R=(1, 0, 0, 1)), rows, un)
    rng = np.random.default_rng(99)
    A, M = 256, 5
    ang = rng.uniform(-np.pi, np.pi, (M, A))
    rr = np.zeros((M, A, 4))
    rr[:, :, 0], rr[:, :, 1] = np.cos(ang) * rng.uniform(0.1, 10, (M, A)), np.sin(ang) * rng.uniform(0.1, 10, (M, A))
    rr[:, :, 2] = rng.normal(0.0, 3.0, (M, A))
    rr[:, :A // 3, 2] += 6.0            # the first third: right-hand sides that mostly admit no input
    un2 = rng.normal(0, 1, (A, 2))
    p = orc.make_params(R=(2.0, 0.3, 0.3, 1.0))
    u2, st2 = orc.qp_batch(p, rr, un2)
    return dict(qp_hand_rows=rows, qp_hand_u_nom=un, qp_hand_u=u, qp_hand_status=st,
                qp_rand_rows=rr, qp_rand_u_nom=un2, qp_rand_u=u2, qp_rand_status=st2)


def build():
    orc.build()
    d = {}
    d.update(barrier_cases())
    d.update(filter_cases())
    d.update(qp_cases())
    return d


if __name__ == "__main__":
    d = build()
    np.savez_compressed(OUT, **d)
    print("%s: %d arrays, %d bytes; %d barrier cases, infeasible QPs %d of %d random / %d of %d hand-made" % (
        OUT, len(d), os.path.getsize(OUT), len(d["bar_kind"]), int((d["qp_rand_status"] == 1).sum()),
        len(d["qp_rand_status"]), int((d["qp_hand_status"] == 1).sum()), len(d["qp_hand_status"])))
