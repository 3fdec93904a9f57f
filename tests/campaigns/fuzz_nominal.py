#!/usr/bin/env python3
"""Randomised campaign for the nominal controller (fp64 against the oracle): random courses
(smooth, self-crossing, with duplicated way-points, 1 ... 6000 points), agents near and far,
arbitrary last target indices.  Exit status 1 on a miss."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402
from sccav_cbf_amd.nominal import BatchedNominal  # noqa: E402

DEV = torch.device("cuda", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    orc.build()
    misses, agents = 0, 0
    for case in range(args.cases):
        N = int(rng.choice([1, 2, 31, 32, 33, 500, 2034, 2048, 2049, 4096, 6000]))
        s = np.linspace(0, rng.uniform(10, 400), N)
        kind = rng.integers(0, 3)
        if kind == 0:      # smooth curve
            x, y = s, 20 * np.sin(s / rng.uniform(10, 60))
        elif kind == 1:    # self-crossing figure, stopping short of closing (way-points that coincide to the last
            # bits but not exactly are ties that np.hypot and squared distances may break differently)
            ph = 0.93 * 2 * np.pi * np.arange(N) / max(N, 1) + 0.1
            x, y = 60 * np.sin(ph), 40 * np.sin(2 * ph + 0.3)
        else:              # noisy polyline with duplicated points
            x, y = s + rng.normal(0, 0.5, N), rng.normal(0, 5, N)
            if N > 10:
                dup = rng.integers(0, N, N // 10)
                x[dup], y[dup] = x[(dup * 7) % N], y[(dup * 7) % N]
        yaw = np.arctan2(np.gradient(y) if N > 1 else np.zeros(1), np.gradient(x) if N > 1 else np.ones(1))
        traj = np.ascontiguousarray(np.stack([x, y, yaw, rng.uniform(3, 12, N)], axis=1))
        A = int(rng.integers(1, 1500))
        at = rng.integers(0, N, A)
        spread = float(rng.choice([0.5, 5.0, 100.0]))
        state = np.ascontiguousarray(np.stack([x[at] + rng.normal(0, spread, A), y[at] + rng.normal(0, spread, A),
                                               rng.uniform(-7, 7, A), rng.uniform(0, 15, A)], axis=1))
        last = rng.integers(0, N, A).astype(np.int32)
        last[rng.uniform(size=A) < 0.3] = 0
        pid = rng.normal(0, 1, (A, 2))
        kw = dict(axle=float(rng.uniform(0, 3)), k=float(rng.uniform(0.1, 2)), ks=float(rng.choice([0.0, 0.01, 1.0])),
                  kp=1.0, ki=0.3, kd=0.05, dt=0.1, target_speed=8.0)
        idx_o, pid_o = last.copy(), pid.copy()
        uo = orc.nominal_batch(state, traj, idx_o, pid_o, **kw)
        ti = torch.from_numpy(last.copy()).to(DEV)
        tp = torch.from_numpy(pid.copy()).to(DEV)
        u = BatchedNominal(**kw).control(torch.from_numpy(state).to(DEV), torch.from_numpy(traj).to(DEV), ti, tp)
        torch.cuda.synchronize()
        agents += A
        di = ti.cpu().numpy() != idx_o
        du = np.abs(u.cpu().numpy() - uo).max(axis=1)
        bad = di | (du > 1e-9) | (np.abs(tp.cpu().numpy() - pid_o).max(axis=1) > 1e-11)
        if bad.any():
            a = int(np.argmax(bad))
            misses += 1
            print("MISS case %d N=%d A=%d kind %d spread %g: %d agents, e.g. %d idx %d/%d du %.3g" % (
                case, N, A, kind, spread, bad.sum(), a, int(ti[a]), idx_o[a], du[a]), flush=True)
    print("cases %d, agents %d, misses %d" % (args.cases, agents, misses))
    sys.exit(1 if misses else 0)


if __name__ == "__main__":
    main()
