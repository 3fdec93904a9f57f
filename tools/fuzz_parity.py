#!/usr/bin/env python3
"""Randomised parity campaign on the GPU: many small batches with random slot kinds (ellipse,
cone, lane, radial, empty), row counts, weights, dynamics, layouts and kernel variants against the
fp64 oracle.  Prints one line per case that misses the bar and a summary; exit status 1 on a miss.

    python tools/fuzz_parity.py [--cases 300] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled  # noqa: E402
from tests import parity  # noqa: E402

DEV = torch.device("cuda", 0)


def random_case(rng, A, M):
    state = np.stack([rng.uniform(-50, 50, A), rng.uniform(0, 30, A), rng.uniform(-3.1, 3.1, A),
                      rng.uniform(0.1, 15, A)], axis=1)
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
        elif k == 2:    # a gently curved lane boundary above or below the agents
            side = rng.choice([-1.0, 1.0])
            slots[m] = np.stack([np.full(A, 15.0 + side * 12.0), rng.uniform(-0.05, 0.05, A), rng.uniform(-1e-3, 1e-3, A),
                                 rng.uniform(-1e-5, 1e-5, A), np.full(A, 1.5), np.zeros(A), np.zeros(A), np.zeros(A)], axis=1)
        elif k == 3:
            slots[m] = np.stack([cx, cy, rng.uniform(1.5, 4, A), rng.uniform(1.5, 4, A), rng.uniform(-3, 3, A),
                                 rng.uniform(-3, 3, A), rng.uniform(0.5, 1.5, A), np.zeros(A)], axis=1)
    obs = np.ascontiguousarray(slots.reshape(M, A, 2, 4).transpose(0, 2, 1, 3))
    return state, u_ref, obs, kinds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--agents", type=int, default=1024)
    ap.add_argument("--only", type=int, default=-1, help="replay the random stream but run (and detail) only this case")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    orc.build()
    misses, total64, total32, st_mism32 = 0, 0, 0, 0
    for case in range(args.cases):
        M = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 16, 17, 24, 32, 48, 64]))
        A = args.agents + int(rng.integers(0, 70))
        state, u_ref, obs, kinds = random_case(rng, A, M)
        q = rng.normal(size=(2, 2))
        R = q @ q.T + 0.3 * np.eye(2)
        alpha = float(rng.choice([0.3, 1.0, 4.0]))
        dyn = str(rng.choice(["dbm", "kbm"]))
        slip = bool(rng.integers(0, 2))
        per_agent = bool(rng.integers(0, 2))
        tiled = bool(rng.integers(0, 2))
        lanes = int(rng.choice([0, 1, 2, 4, 8, 16, 64]))
        if lanes not in (0, 64) and M > 16 * lanes:
            lanes = 0
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
        p = orc.make_params(alpha=alpha, R=R.reshape(-1), flags=flags, dynamics=0 if dyn == "dbm" else 1)
        uo, so, ro = orc.solve_batch(p, state, u_ref, obs, ty, want_rows=True)
        tag = "case %d M=%d A=%d %s slip=%d per_agent=%d tiled=%d lanes=%d alpha=%g" % (
            case, M, A, dyn, slip, per_agent, tiled, lanes, alpha)
        for dtype in (torch.float64, torch.float32):
            f = BatchedCBF(alpha=alpha, R=R, dynamics=dyn, slip_io=slip, lanes_per_qp=lanes)
            tobs = torch.from_numpy(obs).to(DEV).to(dtype)
            if tiled:
                tobs = planes_to_tiled(tobs)
            u, st, rg = f.solve(torch.from_numpy(state).to(DEV).to(dtype), torch.from_numpy(u_ref).to(DEV).to(dtype),
                                tobs, torch.from_numpy(ty).to(DEV), want_rows=True)
            torch.cuda.synchronize()
            u, st, rg = u.double().cpu().numpy(), st.cpu().numpy(), rg.double().cpu().numpy()
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
            well = np.abs(uo[:, 1]) < 1.2 if not slip else np.ones(A, bool)   # tan(delta) near pi/2
            if dtype == torch.float64:
                total64 += A
                bad = (st != so)
                du = (np.abs(u - uo).max(axis=1) / scale)
                bad |= (so == 0) & well & (du > 1e-6)
                if bad.any():
                    misses += 1
                    a = int(np.argmax(bad))
                    print("MISS f64 %s: %d agents, e.g. agent %d status %d/%d du %.3g" % (tag, bad.sum(), a, st[a], so[a], du[a]),
                          flush=True)
            else:
                total32 += A
                st_mism32 += int((st != so).sum())
                # a verdict may differ where the optimum (fp64's or fp32's) lies 1e3 and more from the nominal
                # input: two nearly parallel rows (for DBM every row without dh/dv has the normal (0, a1)), whose
                # vertex fp32 cannot place; anything else is a miss
                far = ((so == 0) & (np.abs(uo - u_ref).max(axis=1) > 1e3)) | ((st == 0) & (np.abs(u - u_ref).max(axis=1) > 1e3))
                unexplained = (st != so) & ~far & ((st == 2) | (so == 2) | slip)
                if unexplained.sum() > max(1, A // 2000):
                    misses += 1
                    print("MISS f32 status %s: %d of %d agents differ without a far vertex" % (tag, unexplained.sum(), A), flush=True)
                if not slip:
                    continue    # conditioning is judged in QP coordinates below
                both = (st == 0) & (so == 0)
                det = parity.active_det(uo, ro)
                # well conditioned: the active pair is not a wedge and the answer is within 50 of the nominal input
                # (a row with a tiny normal puts it far away and scales the row's rounding by the same factor)
                ok = both & (det >= 1e-2) & (np.abs(uo - u_ref).max(axis=1) <= 50)
                du = (np.abs(u - uo).max(axis=1) / scale)
                # 5e-3: a cone row whose normal is the small remainder of cancelling terms (a few 1e-2 against
                # products of several hundred) carries 1e-3 .. 4e-3 relative error in fp32, and so does its answer
                if ok.any() and du[ok].max() > 5e-3:
                    misses += 1
                    a = int(np.argmax(np.where(ok, du, 0)))
                    print("MISS f32 du %s: agent %d du %.3g det %.3g" % (tag, a, du[a], det[a]), flush=True)
        if case % 50 == 49:
            print("... %d cases done, %d misses" % (case + 1, misses), flush=True)
    print("cases %d: fp64 agents %d, fp32 agents %d (status differs on %d = %.2e), misses %d"
          % (args.cases, total64, total32, st_mism32, st_mism32 / max(total32, 1), misses))
    sys.exit(1 if misses else 0)


if __name__ == "__main__":
    main()
