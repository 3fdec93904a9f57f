#!/usr/bin/env python3
"""The bench workload (65536 agents x 16 prepared slots) in fp64 over lanes per QP, one launch at a time and with
three independent launches in flight.  python tools/bench_fp64.py [--lanes 2,4,8,16]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", default="0,2,4,8,16")
    ap.add_argument("--agents", type=int, default=65536)
    ap.add_argument("--eager", type=int, default=0, help="only this many eager launches per lanes value (for counter runs)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    A, n = args.agents, 5
    w = scenarios.prepared(scenarios.config4(A=A, nominal="uniform", seed=1238))
    ty = torch.from_numpy(w["obs_type"]).to(dev)
    ring = []
    for b in range(n):
        sh = 64 * 131 * (b + 1)
        ring.append(dict(state=torch.from_numpy(np.roll(w["state"], sh, axis=0)).to(dev),
                         u_ref=torch.from_numpy(np.roll(w["u_ref"], sh, axis=0)).to(dev),
                         obs=planes_to_tiled(torch.from_numpy(np.roll(w["obs"], sh, axis=2)).to(dev)),
                         u_out=torch.empty((A, 2), dtype=torch.float64, device=dev),
                         status=torch.empty((A,), dtype=torch.int32, device=dev)))
    st = torch.cuda.Stream(dev)
    ab = bench.algorithmic_bytes(A, 16, w=8)
    with torch.cuda.stream(st):
        for lanes in [int(x) for x in args.lanes.split(",")]:
            f = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), streamed_obs=True, lanes_per_qp=lanes)

            def launch(b):
                return lambda: f.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=b["u_out"], status=b["status"])
            steps = [launch(b) for b in ring] * 2
            if args.eager:
                for i in range(args.eager):
                    steps[i % len(steps)]()
                torch.cuda.synchronize()
                continue
            ms1 = bench._time_graph(steps, st, 2 * n, 30)
            ms3 = bench._time_graph(steps, st, 2 * n, 30, branches=3)
            print("fp64 lanes %2d: one at a time %.2f us (%.3f of 8 TB/s), three in flight %.2f us (%.3f)" %
                  (lanes, ms1 * 1e3, ab / (ms1 * 1e-3) / 8e12, ms3 * 1e3, ab / (ms3 * 1e-3) / 8e12), flush=True)


if __name__ == "__main__":
    main()
