#!/usr/bin/env python3
"""Randomised parity campaign on the GPU: many small batches with random slot kinds (ellipse,
cone, lane, radial, empty), row counts, weights, dynamics, layouts and kernel variants against the
fp64 oracle.  Prints one line per case that misses the bar and a summary; exit status 1 on a miss.

    python tests/campaigns/fuzz_parity.py [--cases 300] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_agents, planes_to_tiled  # noqa: E402
from tests import parity  # noqa: E402
from tests.test_gpu_parity import _near_feasibility_boundary  # noqa: E402

DEV = torch.device("cuda", 0)


def random_case(rng, A, M, special_p=0.0):
    state = np.stack([rng.uniform(-50, 50, A), rng.uniform(0, 30, A), rng.uniform(-3.1, 3.1, A),
                      rng.uniform(0.1, 15, A)], axis=1)
    state[rng.uniform(size=A) < 0.05, 3] = 0.0       # at rest: rows without dh/dv lose their normal altogether
    special = np.zeros(A, dtype=bool)
    if special_p > 0:
        # special values, which uniform draws all but never produce: headings on and within 1e-7 ... 1e-3 of the axes
        # (either side), and further down cones that move with the agent's own velocity to 1e-6 ... 1e-2 and stand
        # with the agent on their rim
        special = rng.uniform(size=A) < special_p
        n = int(special.sum())
        base = rng.choice([0.0, 0.5 * np.pi, -0.5 * np.pi, np.pi, -np.pi], n)
        tiny = np.where(rng.uniform(size=n) < 0.2, 0.0, 10.0 ** rng.uniform(-7, -3, n)) * rng.choice([-1.0, 1.0], n)
        th = base + tiny
        th = np.where(th > np.pi, 2 * np.pi - th, np.where(th < -np.pi, -2 * np.pi - th, th))   # stay inside [-pi, pi]
        state[special, 2] = th
    u_ref = np.stack([rng.uniform(-2, 2, A), rng.uniform(-0.5, 0.5, A)], axis=1)
    kinds = rng.choice([0, 1, 2, 3, 255], size=M, p=[0.3, 0.3, 0.1, 0.2, 0.1]).astype(np.uint8)
    slots = np.zeros((M, A, 8))
    for m, k in enumerate(kinds):
        rg, br = rng.uniform(8, 60, A), rng.uniform(-np.pi, np.pi, A)
        cx, cy = state[:, 0] + rg * np.cos(br), state[:, 1] + rg * np.sin(br)
        if k == 0:
            slots[m] = np.stack([cx, cy, rng.uniform(2, 20, A), rng.uniform(1, 10, A), rng.uniform(-3.1, 3.1, A),
                                 rng.uniform(-3, 3, A), rng.uniform(-3, 3, A), np.zeros(A)], axis=1)
        elif k == 1:
            slots[m] = np.stack([cx, cy, rng.uniform(-3.1, 3.1, A), rng.uniform(0, 6, A), rng.uniform(2, 7, A),
                                 rng.choice([0.0, 0.1, -0.2], A), np.zeros(A), np.zeros(A)], axis=1)
            if special.any():
                n = int(special.sum())
                sl = slots[m][special]
                sl[:, 2] = state[special, 2]                                                   # the agent's heading,
                sl[:, 3] = state[special, 3] * (1.0 + rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-6, -2, n))  # its speed
                sl[:, 5] = 0.0
                rim = rng.uniform(size=n) < 0.5                                                # and half of them on the rim
                d = np.where(rim, sl[:, 4] * (1.0 + rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-7, -3, n)), rg[special])
                sl[:, 0] = state[special, 0] + d * np.cos(br[special])
                sl[:, 1] = state[special, 1] + d * np.sin(br[special])
                slots[m][special] = sl
        elif k == 2:    # a gently curved lane boundary above or below the agents
            side = rng.choice([-1.0, 1.0])
            slots[m] = np.stack([np.full(A, 15.0 + side * 12.0), rng.uniform(-0.05, 0.05, A), rng.uniform(-1e-3, 1e-3, A),
                                 rng.uniform(-1e-5, 1e-5, A), np.full(A, 1.5), np.zeros(A), np.zeros(A), np.zeros(A)], axis=1)
        elif k == 3:
            slots[m] = np.stack([cx, cy, rng.uniform(1.5, 4, A), rng.uniform(1.5, 4, A), rng.uniform(-3, 3, A),
                                 rng.uniform(-3, 3, A), rng.uniform(0.5, 1.5, A), np.zeros(A)], axis=1)
    obs = np.ascontiguousarray(slots.reshape(M, A, 2, 4).transpose(0, 2, 1, 3))
    # half of the ellipse and cone slots in their prepared forms (kinds 4 and 5: same obstacle, constants evaluated)
    from sccav_cbf_amd import scenarios
    wp = scenarios.prepared(dict(obs=obs, obs_type=kinds))
    take = (rng.uniform(size=M) < 0.5) & ((kinds == 0) | (kinds == 1))
    obs[take] = wp["obs"][take]
    kinds = np.where(take, wp["obs_type"], kinds).astype(np.uint8)
    # a fifth of the slots as bounding boxes (kinds 6, 7: x, y, extent x, y | yaw, speed, buffer): the filter makes the
    # ellipse / cone of from_bounding_box out of them
    for m in range(M):
        if kinds[m] != 255 and rng.uniform() < 0.2:
            rg, br = rng.uniform(8, 60, A), rng.uniform(-np.pi, np.pi, A)
            box = np.stack([state[:, 0] + rg * np.cos(br), state[:, 1] + rg * np.sin(br), rng.uniform(1.5, 4, A),
                            rng.uniform(0.7, 2, A), rng.uniform(-3.1, 3.1, A), rng.uniform(0, 8, A),
                            np.full(A, rng.uniform(0, 1.5)), np.zeros(A)], axis=1)
            obs[m] = box.reshape(A, 2, 4).transpose(1, 0, 2)
            kinds[m] = rng.choice([6, 7])
    return state, u_ref, np.ascontiguousarray(obs), kinds


def inscribed_radius(rows):
    """Radius of the largest circle inside the half-planes a . u >= b of rows (M, >= 3), capped at 1; negative when
    they have no common point (then the least distance by which they miss one)."""
    from scipy.optimize import linprog
    a, b = rows[:, :2], rows[:, 2]
    n = np.hypot(a[:, 0], a[:, 1])
    keep = n > 0
    res = linprog(c=[0.0, 0.0, -1.0], A_ub=np.c_[-a[keep], n[keep]], b_ub=-b[keep],
                  bounds=[(None, None), (None, None), (None, 1.0)], method="highs")
    return float(res.x[2]) if res.status == 0 else float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--agents", type=int, default=1024)
    ap.add_argument("--only", type=int, default=-1, help="replay the random stream but run (and detail) only this case")
    ap.add_argument("--special", type=float, default=0.0,
                    help="share of the agents given special values (headings on the axes and next to them, cones at the "
                         "agent's own velocity and with the agent on their rim); 0 keeps the random stream of earlier runs")
    ap.add_argument("--extra", type=float, default=0.0,
                    help="share of the slots replaced by lanes of degree 4 ... 6 (kind 8); with it the layout is drawn "
                         "from planes / tiles / per-agent lists and 513 rows join the row counts.  Drawn from a stream "
                         "of its own: 0 keeps the random stream of earlier runs")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    xrng = np.random.default_rng([args.seed, 7])
    orc.build()
    misses, total64, total32, st_mism32, worst32 = 0, 0, 0, 0, 0.0
    for case in range(args.cases):
        M = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 16, 17, 24, 32, 48, 64, 100, 257]))
        if args.extra > 0 and xrng.uniform() < 0.03:
            M = 513
        A = (args.agents if M <= 64 else 128) + int(rng.integers(0, 70))
        state, u_ref, obs, kinds = random_case(rng, A, M, args.special)
        layout = -1
        if args.extra > 0:
            layout = int(xrng.integers(0, 3))       # planes, tiles, per-agent lists
            for m in range(M):
                if xrng.uniform() < args.extra:     # a lane boundary of degree 4 ... 6: c0 ... c6, buffer
                    deg = int(xrng.integers(4, 7))
                    c = np.stack([np.full(A, 15.0 + xrng.choice([-1.0, 1.0]) * 12.0), xrng.uniform(-0.05, 0.05, A),
                                  xrng.uniform(-1e-3, 1e-3, A), xrng.uniform(-1e-5, 1e-5, A), xrng.uniform(-1e-7, 1e-7, A),
                                  xrng.uniform(-1e-9, 1e-9, A), xrng.uniform(-1e-11, 1e-11, A), np.full(A, 1.5)], axis=1)
                    c[:, deg + 1:7] = 0.0
                    obs[m] = c.reshape(A, 2, 4).transpose(1, 0, 2)
                    kinds[m] = 8
        q = rng.normal(size=(2, 2))
        R = q @ q.T + 0.3 * np.eye(2)
        alpha = float(rng.choice([0.3, 1.0, 4.0]))
        dyn = str(rng.choice(["dbm", "kbm", "dum", "sadbm"]))
        slip = bool(rng.integers(0, 2))
        per_agent = bool(rng.integers(0, 2))
        tiled = bool(rng.integers(0, 2))
        lanes = int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64]))
        if lanes not in (0, 16, 32, 64) and M > 16 * lanes:
            lanes = 0
        if lanes == 16 and M > 400:     # sixteen tables of so many fp64 rows do not fit a CU's LDS
            lanes = 0
        if dyn in ("dum", "sadbm"):     # these models run with 4 lanes per QP (up to 64 rows), 8 (128), 16, 32 or 64
            lanes = lanes if lanes in (0, 16, 32, 64) or (lanes == 4 and M <= 64) or (lanes == 8 and M <= 128) else 0
        sa_dt = float(rng.choice([0.02, 0.1]))
        sa_slip = np.stack([rng.uniform(-0.3, 0.3, A), rng.uniform(-0.3, 0.3, A)], axis=1)
        cone_beta = bool(rng.integers(0, 2))
        if dyn == "kbm":
            u_ref[:, 0] = np.abs(u_ref[:, 0]) + 2.0
        ty = kinds
        flags = orc.FLAG_SLIP_IO if slip else 0
        if per_agent:
            ty = np.repeat(kinds[:, None], A, axis=1)
            ty[rng.uniform(size=ty.shape) < 0.2] = 255
            ty = np.ascontiguousarray(ty)
            flags |= orc.FLAG_TYPE_PER_AGENT
        if args.only >= 0 and case != args.only:
            continue
        dcode = {"dbm": 0, "kbm": 1, "dum": 2, "sadbm": 3}[dyn]
        if dyn in ("dum", "sadbm"):
            slip = dyn == "dum"          # DUM has no steering maps; SADBM is judged in QP coordinates below
            flags &= ~orc.FLAG_SLIP_IO
        if dyn == "sadbm" and cone_beta:
            flags |= orc.FLAG_CONE_BETA_STATE
        p = orc.make_params(alpha=alpha, R=R.reshape(-1), flags=flags, dynamics=dcode)
        u_given = u_ref
        if dyn == "sadbm":
            slo = sa_slip.copy()
            uo, so, ro = orc.solve_sa_batch(p, sa_dt, state, slo, u_given, obs, ty, want_rows=True)
            # QP coordinates [a, d(beta)/dt]: nominal input and answers
            beta_ref = np.arctan2(1.45 * np.tan(u_given[:, 1]), 2.9)
            u_ref = np.stack([u_given[:, 0], (beta_ref - sa_slip[:, 1]) / sa_dt], axis=1)
            uo = np.stack([uo[:, 0], (slo[:, 0] - sa_slip[:, 0]) / sa_dt], axis=1)
            uo[so == 2] = u_ref[so == 2]
        else:
            uo, so, ro = orc.solve_batch(p, state, u_ref, obs, ty, want_rows=True)
        # the rows of the same data rounded to fp32 (what the fp32 kernels are given): how far rounding the
        # inputs alone moves a row measures how ill-conditioned that row is in single precision
        r32 = lambda v: v.astype(np.float32).astype(np.float64)       # noqa: E731

        def rows_of(rnd):
            if dyn == "sadbm":
                return orc.solve_sa_batch(p, sa_dt, rnd(state), rnd(sa_slip), rnd(u_given), rnd(obs), ty, want_rows=True)[2]
            return orc.solve_batch(p, rnd(state), rnd(u_given), rnd(obs), ty, want_rows=True)[2]

        jrng = np.random.default_rng([args.seed, case])   # its own stream: --only replays a case exactly

        def jitter(v):      # every value one fp32 ulp up or down at random
            f = v.astype(np.float32)
            return np.where(jrng.integers(0, 2, f.shape) == 1, np.nextafter(f, np.float32(np.inf)),
                            np.nextafter(f, np.float32(-np.inf))).astype(np.float64)
        ro32 = rows_of(r32)
        sens = np.abs(ro - ro32)
        for _ in range(7):
            sens = np.maximum(sens, np.abs(rows_of(jitter) - ro32))
        if layout >= 0:
            tiled = layout == 1
        tag = "case %d M=%d A=%d %s slip=%d per_agent=%d tiled=%d lists=%d lanes=%d alpha=%g" % (
            case, M, A, dyn, slip, per_agent, tiled, layout == 2, lanes, alpha)
        for dtype in (torch.float64, torch.float32):
            f = BatchedCBF(alpha=alpha, R=R, dynamics=dyn, slip_io=slip and dyn in ("dbm", "kbm"), lanes_per_qp=lanes)
            tobs = torch.from_numpy(obs).to(DEV).to(dtype)
            if tiled:
                tobs = planes_to_tiled(tobs)
            elif layout == 2:
                tobs = planes_to_agents(tobs)
            kw = {}
            if dyn == "sadbm":
                kw = dict(slip=torch.from_numpy(sa_slip).to(DEV).to(dtype), dt=sa_dt, cone_beta_state=cone_beta)
            u, st, rg = f.solve(torch.from_numpy(state).to(DEV).to(dtype), torch.from_numpy(u_given).to(DEV).to(dtype),
                                tobs, torch.from_numpy(ty).to(DEV), want_rows=True, **kw)
            torch.cuda.synchronize()
            u, st, rg = u.double().cpu().numpy(), st.cpu().numpy(), rg.double().cpu().numpy()
            if dyn == "sadbm":
                slg = kw["slip"].double().cpu().numpy()
                if dtype == torch.float64 and (np.abs(slg - slo).max() > 1e-7 * (1 + np.abs(slo).max())):
                    misses += 1
                    print("MISS f64 slip state %s: %.3g" % (tag, np.abs(slg - slo).max()), flush=True)
                u = np.stack([u[:, 0], (slg[:, 0] - sa_slip[:, 0]) / sa_dt], axis=1)
                u[st == 2] = u_ref[st == 2]
            if args.only >= 0 and dtype == torch.float32:
                np.set_printoptions(linewidth=200, precision=6)
                scale = np.maximum(1.0, np.hypot(uo[:, 0], uo[:, 1]))
                du = np.abs(u - uo).max(axis=1) / scale
                worst = list(np.where(st != so)[0][:4]) + list(np.argsort(np.where((st == 0) & (so == 0), du, 0))[::-1][:3])
                for a in worst:
                    print("agent %d: status gpu %d oracle %d, u gpu %s oracle %s, u_ref %s, state %s" % (a, st[a], so[a], u[a], uo[a], u_ref[a], state[a]))
                    kk = ty[:, a] if per_agent else ty
                    rel = np.abs(rg[:, a, :] - ro[:, a, :]).max(axis=1) / (np.abs(ro[:, a, :]).max(axis=1) + 1)
                    for m in np.argsort(rel)[::-1][:3]:
                        print("    slot %d kind %d rel row error %.3g: gpu %s oracle %s slot %s" % (m, kk[m], rel[m], rg[m, a], ro[m, a], obs[m, :, a, :].reshape(-1)))
            scale = np.maximum(1.0, np.hypot(uo[:, 0], uo[:, 1]))
            well = np.abs(uo[:, 1]) < 1.2 if not slip and dyn != "sadbm" else np.ones(A, bool)   # tan(delta) near pi/2
            if dtype == torch.float64:
                total64 += A
                bad = (st != so)
                # a verdict may differ where the feasible set is a point: planted special values make three rows meet
                # to 1e-11 (largest circle inside the rows' half-planes, by a linear program: radius below 1e-9)
                point = np.zeros(A, dtype=bool)
                for a_ in np.where(bad & (st != 2) & (so != 2))[0]:
                    point[a_] = abs(inscribed_radius(ro[:, a_, :])) < 1e-9
                du = (np.abs(u - uo).max(axis=1) / scale)
                # in steering coordinates a slip angle of thousands of radians (a vertex far away on nearly
                # parallel rows) comes back through tan / atan2: judged only where the answer is near the input
                near = np.abs(uo[:, 0] - u_ref[:, 0]) <= 1e2
                if not slip and dyn in ("dbm", "kbm"):
                    un_int = u_ref.copy()
                    un_int[:, 1] = (np.arctan2(1.45 * np.tan(u_ref[:, 1]), 2.9) if dyn == "dbm"
                                    else u_ref[:, 0] * np.tan(u_ref[:, 1]) / 2.9)
                    ui, _ = orc.qp_batch(orc.make_params(R=R.reshape(-1)), np.ascontiguousarray(ro), un_int)
                    near &= np.abs(ui - un_int).max(axis=1) <= 1e2
                bad |= (so == 0) & well & near & (du > 1e-6)
                bad &= ~point
                if bad.any():
                    misses += 1
                    a = int(np.argmax(bad))
                    print("MISS f64 %s: %d agents, e.g. agent %d status %d/%d du %.3g" % (tag, bad.sum(), a, st[a], so[a], du[a]),
                          flush=True)
                    if args.only >= 0:
                        np.set_printoptions(linewidth=200, precision=12)
                        print("  u gpu %s oracle %s u_ref %s state %s" % (u[a], uo[a], u_ref[a], state[a]))
                        lhs = ro[:, a, 0] * uo[a, 0] + ro[:, a, 1] * np.arctan2(1.45 * np.tan(uo[a, 1]), 2.9)
                        print("  oracle rows (a0, a1, b, h) and slack at the oracle's answer (beta coordinates):")
                        for m in range(M):
                            print("   ", ro[m, a], lhs[m] - ro[m, a, 2], " gpu row", rg[m, a])
            else:
                total32 += A
                st_mism32 += int((st != so).sum())
                # stage 1, rows: every element against the magnitude of its row; allowed: 1e-3, or 16 times
                # what one fp32 ulp on the inputs does to the exact row (largest of eight trials: a cone seen at
                # nearly the agent's own velocity divides by the small relative speed)
                rscale = np.abs(ro32).max(axis=2, keepdims=True) + 1.0
                relm = (np.abs(rg - ro32) / rscale).max(axis=2) - 16.0 * (sens / rscale).max(axis=2)
                rel = relm.max()
                if rel > 1e-3:
                    misses += 1
                    m_, a_ = np.unravel_index(np.argmax(relm), relm.shape)
                    print("MISS f32 rows %s: %.3g at slot %d (kind %d) agent %d\n   gpu %s\n   oracle %s\n   slot %s state %s"
                          % (tag, rel, m_, (ty[m_, a_] if per_agent else ty[m_]), a_, rg[m_, a_], ro[m_, a_],
                             obs[m_, :, a_, :].reshape(-1), state[a_]), flush=True)
                if ((st == 2) != (so == 2)).any():
                    misses += 1
                    print("MISS f32 non-finite verdicts differ %s" % tag, flush=True)
                if not slip and dyn != "sadbm":
                    continue    # the QP stage is judged in QP coordinates
                # stage 2, the QP on the rows the GPU itself formed, against the exact optimum of those rows.
                # A verdict may differ where the optimum (either one) lies 1e3 and more from the nominal input
                # (two nearly parallel rows: for DBM every row without dh/dv has the normal (0, a1)) or where
                # the rows are within 1e-3 (relative) of the boundary of feasibility.
                un = u_ref.astype(np.float32).astype(np.float64)
                pq = orc.make_params(R=R.reshape(-1))
                rq = np.ascontiguousarray(rg)
                uq, sq = orc.qp_batch(pq, rq, un)
                far = ((sq == 0) & (np.abs(uq - un).max(axis=1) > 1e3)) | ((st == 0) & (np.abs(u - un).max(axis=1) > 1e3))
                idx = np.where((st != sq) & (st != 2) & ~far)[0]
                # ... or where the optimum (the exact one, or the GPU's) sits on two rows that single precision cannot
                # tell from parallel (|sin| < 1e-5; the kernel's own threshold is 1e-6): facing each other such rows
                # leave a wedge a few ulp wide, which exists or not by the last bits
                wedge = np.where(sq[idx] == 0, parity.active_det(uq, rq)[idx], parity.active_det(u, rq)[idx]) < 1e-5
                idx = idx[~wedge]
                if idx.size and not all(_near_feasibility_boundary(idx[k:k + 1], sq, rq, un, R) for k in range(idx.size)):
                    misses += 1
                    print("MISS f32 status %s: %d of %d agents differ from the exact QP on the same rows without a far "
                          "vertex or a feasibility margin below 1e-3" % (tag, idx.size, A), flush=True)
                    if args.only >= 0:
                        np.set_printoptions(linewidth=200, precision=9)
                        for a_ in idx:
                            print("  agent %d status gpu %d exact %d oracle %d; u gpu %s exact %s nominal %s" % (
                                a_, st[a_], sq[a_], so[a_], u[a_], uq[a_], un[a_]))
                            for m_ in range(M):
                                print("     row", rq[m_, a_], " oracle row", ro[m_, a_])
                both = (st == 0) & (sq == 0)
                det = parity.active_det(uq, rq)
                ok = both & (det >= 1e-2) & (np.abs(uq - un).max(axis=1) <= 50)
                uscale = np.maximum(1.0, np.hypot(uq[:, 0], uq[:, 1]))
                dq = np.abs(u - uq).max(axis=1) / uscale
                # a row enters the QP as its distance from the nominal input, (b - a . u_nom) / |a|: where b and
                # a . u_nom nearly cancel, an fp32 ulp of either (4 of them allowed) is an error of that distance, and a
                # vertex moves by it over |sin|
                nrm = np.hypot(rq[:, :, 0], rq[:, :, 1]) + 1e-300
                canc = ((np.abs(rq[:, :, 2]) + np.abs(rq[:, :, 0] * un[None, :, 0]) + np.abs(rq[:, :, 1] * un[None, :, 1])) / nrm).max(axis=0)
                tolq = 1e-3 + 4.0 * 6e-8 * canc / (det * uscale)
                if ok.any() and (dq[ok] > tolq[ok]).any():
                    misses += 1
                    a = int(np.argmax(np.where(ok, dq - tolq, -np.inf)))
                    print("MISS f32 qp %s: agent %d du %.3g (allowed %.3g) det %.3g" % (tag, a, dq[a], tolq[a], det[a]), flush=True)
                # for the record: distance to the oracle's answer among well-conditioned optima
                both = (st == 0) & (so == 0)
                ok = both & (parity.active_det(uo, ro) >= 1e-2) & (np.abs(uo - u_ref).max(axis=1) <= 50)
                if ok.any():
                    worst32 = max(worst32, float((np.abs(u - uo).max(axis=1) / scale)[ok].max()))
        if case % 50 == 49:
            print("... %d cases done, %d misses" % (case + 1, misses), flush=True)
    print("cases %d: fp64 agents %d, fp32 agents %d (status differs on %d = %.2e; worst |du| to the oracle among "
          "well-conditioned optima %.3g), misses %d"
          % (args.cases, total64, total32, st_mism32, st_mism32 / max(total32, 1), worst32, misses))
    sys.exit(1 if misses else 0)


if __name__ == "__main__":
    main()
