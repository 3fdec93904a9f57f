#!/usr/bin/env python3
"""Latency of the small configurations: C2 (1 ego x 32 radial obstacles, fp64) and C3 (4096 x 8 ellipses, fp32)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF  # noqa: E402

dev = torch.device("cuda", 0)


def graph_time(f, w, dtype, reps=200):
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in w.items()}
    st, ur, ob = t["state"].to(dtype), t["u_ref"].to(dtype), t["obs"].to(dtype).contiguous()
    A = st.shape[0]
    u = torch.empty((A, 2), dtype=dtype, device=dev)
    s = torch.empty((A,), dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        f.solve(st, ur, ob, t["obs_type"], u_out=u, status=s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(10):
                f.solve(st, ur, ob, t["obs_type"], u_out=u, status=s)
        for _ in range(20):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            g.replay()
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * 10)
        # one eager call with a host round trip, as a per-tick caller would see it
        t0 = time.perf_counter()
        for _ in range(200):
            f.solve(st, ur, ob, t["obs_type"], u_out=u, status=s)
            u.cpu()
        eager = (time.perf_counter() - t0) / 200 * 1e6
    return us, eager


w2 = scenarios.radial_seekers(M=32, A=1, seed=1236)
us, eager = graph_time(BatchedCBF(alpha=1.0, R=np.eye(2)), w2, torch.float64)
print("C2  1 ego x 32 radial rows, fp64: %.2f us per launch in a graph, %.1f us per eager call + copy back" % (us, eager))
w3 = scenarios.config3(A=4096)
us, eager = graph_time(BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5])), w3, torch.float32)
print("C3  4096 x 8 ellipse rows, fp32: %.2f us per launch in a graph (%.3e solves/s), %.1f us eager + copy back"
      % (us, 4096 / (us * 1e-6), eager))
w5 = scenarios.lanes_and_cones(A=16384)
for dt in (torch.float32, torch.float64):
    us, eager = graph_time(BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5])), w5, dt)
    print("C5  16384 x (2 lanes + 6 cones), %s: %.2f us per launch in a graph (%.3e solves/s)"
          % ("fp32" if dt == torch.float32 else "fp64", us, 16384 / (us * 1e-6)))
