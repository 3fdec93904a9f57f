#!/usr/bin/env python3
"""Launch times of the ingestion kernels (boxes -> slots, lane points -> lane slots) on one GPU:
hipGraph of 10 launches, events around 200 replays.  python tools/bench_ingest.py [--agents A]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import ingest  # noqa: E402


def timed(fn, reps=200, per_graph=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(per_graph):
                fn()
        for _ in range(50):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * per_graph) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=65536)
    args = ap.parse_args()
    A, M = args.agents, 16
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    for ty in (torch.float32, torch.float64):
        obs = torch.zeros((-(-A // 64), M, 2, 64, 4), dtype=ty, device=dev)
        for K in (4, 8, 16, 32, 48, 64, 96):
            x = np.sort(rng.uniform(-5.0, 60.0, (K, A)), axis=0)
            y = 0.002 * x * x - 0.05 * x + 3.0 + rng.normal(0.0, 0.05, (K, A))
            px, py = torch.tensor(x, dtype=ty, device=dev), torch.tensor(y, dtype=ty, device=dev)
            for deg in (1, 3):
                if K <= deg:
                    continue
                us = timed(lambda: ingest.lane_fit(px, py, obs, 0, degree=deg))
                print("lane_fit %s K=%2d degree=%d  %7.2f us  (%.1f GB/s of points)" % (
                    str(ty)[6:], K, deg, us, 2 * K * A * px.element_size() / us / 1e3), flush=True)
        box = torch.tensor(rng.uniform(0.5, 3.0, (M, 2, A, 4)), dtype=ty, device=dev)
        kind = torch.tensor([0, 1] * (M // 2), dtype=torch.uint8, device=dev)
        us = timed(lambda: ingest.ingest_bbox(box, kind, 1.5, True, obs))
        print("ingest_bbox %s M=%d  %7.2f us" % (str(ty)[6:], M, us), flush=True)

if __name__ == "__main__":
    main()
