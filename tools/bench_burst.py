#!/usr/bin/env python3
"""A short burst from an idle GPU: K = 20 steps of the bench workload between two synchronizations (the shape of
`bench.py --steps 20`), by wall clock, median of 50 bursts: (a) one replay of a 20-launch graph, serial; (b) the same
graph with three parallel branches; (c) three single-stream graphs of 7 + 7 + 6 launches started on three streams.
python tools/bench_burst.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, capture_independent  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    K = 20
    ring = bench.make_ring(0, dev, 10, layout="tiles", slots="prepared")
    flt = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), streamed_obs=True)
    ty = ring[0]["obs_type"]

    def step(i):
        b = ring[i % len(ring)]
        return lambda: flt.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=b["u_out"], status=b["status"])
    steps = [step(i) for i in range(K)]
    for f in steps:
        f()
    torch.cuda.synchronize()
    main_s = torch.cuda.Stream(dev)
    side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    with torch.cuda.stream(main_s):
        g1 = capture_independent(steps, 1)
        g3 = capture_independent(steps, 3)
    parts = []
    for j, s in enumerate([main_s] + side):
        with torch.cuda.stream(s):
            parts.append((s, capture_independent(steps[j::3], 1)))

    def burst_a():
        with torch.cuda.stream(main_s):
            g1.replay()

    def burst_b():
        with torch.cuda.stream(main_s):
            g3.replay()

    def burst_c():
        for s, g in parts:
            with torch.cuda.stream(s):
                g.replay()

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def burst_d():   # (b) between two timing events, as a profiled region has them
        with torch.cuda.stream(main_s):
            e0.record(main_s)
            g3.replay()
            e1.record(main_s)

    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    made = {}
    for label, flags in (("device-scope release", 0x40000000), ("no system fence", 0x20000000)):
        h0, h1 = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(h0), flags) == 0
        assert hip.hipEventCreateWithFlags(ctypes.byref(h1), flags) == 0
        made[label] = (h0, h1)

    def with_events(label):
        h0, h1 = made[label]
        sp = ctypes.c_void_p(main_s.cuda_stream)

        def fn():
            with torch.cuda.stream(main_s):
                hip.hipEventRecord(h0, sp)
                g3.replay()
                hip.hipEventRecord(h1, sp)
        return fn

    def branches(nb):
        with torch.cuda.stream(main_s):
            g = capture_independent(steps, nb)

        def fn():
            with torch.cuda.stream(main_s):
                g.replay()
        return fn

    cases = [("one serial graph", burst_a), ("three branches + events, device-scope release", with_events("device-scope release")),
             ("three branches + events, no system fence", with_events("no system fence")), ("one graph, three branches", burst_b),
             ("three graphs on three streams", burst_c), ("three branches + two event records", burst_d)]
    cases += [("one graph, %d branches" % nb, branches(nb)) for nb in (2, 4, 5, 7, 10)]
    for name, fn in cases:
        # keep the clocks up between the bursts, then idle the GPU as a synchronize does
        ts, th = [], []
        for rep in range(60):
            for _ in range(30):
                burst_a()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            th.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts, th = np.array(ts[10:]) * 1e6, np.array(th[10:]) * 1e6
        print("%-46s %6.1f us per burst of %d steps (median; min %.1f) = %.2f us per step; the host call returns after %.1f us" %
              (name, np.median(ts), K, ts.min(), np.median(ts) / K, np.median(th)), flush=True)


if __name__ == "__main__":
    main()
