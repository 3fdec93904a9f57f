#!/usr/bin/env python3
"""Do independent launches overlap when a captured graph has parallel branches?  The bench workload (65536 agents x 16
rows fp32, ring of 10 distinct batches), 100 launches per graph, dealt to 1, 2, 3 or 4 branches (side streams forked
from and joined to the capture stream).  python tools/bench_overlap.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ring = bench.make_ring(0, dev, 10, layout="tiles", slots="prepared")
    flt = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), streamed_obs=True)
    ty = ring[0]["obs_type"]

    def step(i):
        b = ring[i % len(ring)]
        flt.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=b["u_out"], status=b["status"])

    for nb in (1, 2, 3, 4):
        main_s = torch.cuda.Stream(dev)
        side = [torch.cuda.Stream(dev) for _ in range(nb - 1)]
        with torch.cuda.stream(main_s):
            for i in range(10):
                step(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main_s):
                for s in side:
                    s.wait_stream(main_s)
                for i in range(100):
                    s = ([main_s] + side)[i % nb]
                    with torch.cuda.stream(s):
                        step(i)
                for s in side:
                    main_s.wait_stream(s)
            for _ in range(100):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            t0 = time.perf_counter()
            for _ in range(200):
                g.replay()
            host = time.perf_counter() - t0
            e1.record(main_s)
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (200 * 100) * 1e3
        print("%d branch(es): %.2f us per launch, %.3f of 8 TB/s; the host's replay calls took %.2f us per launch" %
              (nb, us, 35913728 / (us * 1e-6) / 8e12, host / (200 * 100) * 1e6), flush=True)

    # serial graphs of 100 launches, one per stream, replayed side by side (no fork / join inside a graph)
    for ns in (2, 3, 4):
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        graphs = []
        for j, s in enumerate(streams):
            with torch.cuda.stream(s):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for i in range(100):
                        step(i * ns + j)
                graphs.append(g)
        def walk(n):
            for _ in range(n):
                for s, g in zip(streams, graphs):
                    with torch.cuda.stream(s):
                        g.replay()
        walk(50)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        walk(100)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / (100 * 100 * ns) * 1e6
        print("%d streams, a serial graph of 100 launches on each: %.2f us per launch by the wall clock, %.3f of 8 TB/s; "
              "the host's replay calls took %.2f us per launch" % (ns, us, 35913728 / (us * 1e-6) / 8e12,
                                                                  host / (100 * 100 * ns) * 1e6), flush=True)


if __name__ == "__main__":
    main()
