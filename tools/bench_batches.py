#!/usr/bin/env python3
"""K batches of the bench workload (65536 agents x 16 rows, tiled prepared slots) in ONE launch
(sccbf_solve_batches_*) against one launch per batch: microseconds per batch and share of 8 TB/s.

    python tools/bench_batches.py [--dtype f32|f64] [--agents 65536] [--rows 16] [--ring 20] [--lanes 0] [--layout tiles|agents|planes]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, BatchTable, planes_to_agents, planes_to_tiled  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--agents", type=int, default=65536)
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--ring", type=int, default=20)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--slots", default="prepared")
    ap.add_argument("--ks", default="1,2,4,8,10,16,20")
    ap.add_argument("--layout", default="tiles", choices=("tiles", "agents", "planes"))
    args = ap.parse_args()
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    w = 4 if args.dtype == "f32" else 8
    dev = torch.device("cuda", 0)
    A, M = args.agents, args.rows
    base = scenarios.monte_carlo(A, (M + 1) // 2, M // 2, 1238, nominal="uniform")
    if args.slots == "prepared":
        base = scenarios.prepared(base)
    ring = []
    for b in range(args.ring):
        sh = 64 * 131 * (b + 1)
        ring.append(dict(state=torch.from_numpy(np.roll(base["state"], sh, axis=0)).to(dev).to(dt),
                         u_ref=torch.from_numpy(np.roll(base["u_ref"], sh, axis=0)).to(dev).to(dt),
                         obs={"tiles": planes_to_tiled, "agents": planes_to_agents, "planes": lambda o: o.contiguous()}[args.layout](
                             torch.from_numpy(np.roll(base["obs"], sh, axis=2)).to(dev).to(dt)),
                         obs_type=torch.from_numpy(base["obs_type"]).to(dev)))
    flt = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), lanes_per_qp=args.lanes, streamed_obs=True)
    table = BatchTable(ring)
    bytes_per = A * (w * (4 + 2 + 8 * M + 2) + 4)
    st = torch.cuda.Stream(dev)
    print("dtype %s, %d agents x %d rows, ring of %d batches (%.0f MB), %.2f MB per batch"
          % (args.dtype, A, M, args.ring, args.ring * bytes_per / 1e6, bytes_per / 1e6))
    with torch.cuda.stream(st):
        def walk_fused(K):
            for f in range(0, args.ring - K + 1, K):
                flt.solve_batches(table, first=f, count=K)
            return (args.ring // K) * K

        def walk_separate():
            for b in table.batches:
                flt.solve(b["state"], b["u_ref"], b["obs"], b["obs_type"], u_out=b["u_out"], status=b["status"])
            return args.ring

        def timed(walk, reps):
            for _ in range(max(3, reps // 5)):
                n = walk()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(st)
            for _ in range(reps):
                walk()
            e1.record(st)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e6 / (reps * n)
            return e0.elapsed_time(e1) * 1e3 / (reps * n), wall

        us, wall = timed(walk_separate, args.reps)
        print("one launch per batch (eager)      %7.2f us per batch by events, %7.2f by the wall clock, %.3f of 8 TB/s"
              % (us, wall, bytes_per / us / 8e6))
        for K in [int(k) for k in args.ks.split(",")]:
            if K > args.ring:
                continue
            us, wall = timed(lambda: walk_fused(K), args.reps)
            print("%2d batches per launch             %7.2f us per batch by events, %7.2f by the wall clock, %.3f of 8 TB/s"
                  % (K, us, wall, bytes_per / us / 8e6))
        # one launch between two synchronisations, as a 20-step timed region sees it
        K = min(20, args.ring)
        ts = []
        for _ in range(50):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            flt.solve_batches(table, first=0, count=K)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6)
        ts = np.sort(ts)
        print("one launch of %d batches between two synchronisations: median %.1f us, min %.1f (%.2f us per batch, %.3f of 8 TB/s)"
              % (K, ts[len(ts) // 2], ts[0], ts[len(ts) // 2] / K, bytes_per * K / ts[len(ts) // 2] / 8e6))


if __name__ == "__main__":
    main()
