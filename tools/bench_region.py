#!/usr/bin/env python3
"""What a twenty-step timed region costs the host beyond its one kernel: the same launch of twenty 65536-agent
batches between two synchronisations, (a) bare with the launch bound once, (b) through solve_batches, (c) bracketed
by two events recorded on the stream.  Median of 60 regions each.  (Handing the runtime a start / stop event pair
with the launch, hipExtLaunchKernel, was measured here too: 146 us against 123 bare and 126 with stream events.)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, BatchTable, planes_to_tiled  # noqa: E402

dev = torch.device("cuda", 0)
base = scenarios.prepared(scenarios.config4(A=65536, nominal="uniform"))
ring = []
for b in range(20):
    sh = 64 * 131 * (b + 1)
    ring.append(dict(state=torch.from_numpy(np.roll(base["state"], sh, axis=0)).float().to(dev),
                     u_ref=torch.from_numpy(np.roll(base["u_ref"], sh, axis=0)).float().to(dev),
                     obs=planes_to_tiled(torch.from_numpy(np.roll(base["obs"], sh, axis=2)).float().to(dev)),
                     obs_type=torch.from_numpy(base["obs_type"]).to(dev)))
flt = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), streamed_obs=True)
table = BatchTable(ring)
st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    full = flt.bind_batches(table)
    for _ in range(300):
        full()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for x in e:
        x.record(st)
    torch.cuda.synchronize()

    def region(kind):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if kind == "events":
            e[2].record(st)
        if kind == "unbound":
            flt.solve_batches(table)
        else:
            full()
        if kind == "events":
            e[3].record(st)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e6
        devt = e[2].elapsed_time(e[3]) * 1e3 if kind == "events" else 0.0
        return wall, devt

    for kind in ("bare", "unbound", "events", "bare"):
        r = np.array([region(kind) for _ in range(60)])
        print("%-8s wall median %.1f us (min %.1f, first %.1f), device %.1f us" % (
            kind, np.median(r[:, 0]), r[:, 0].min(), r[0, 0], np.median(r[:, 1])))
    # the same after the GPU has idled for a while, as after the set-up of a timed region
    for idle_ms in (0.2, 2.0, 20.0):
        r = []
        for _ in range(20):
            torch.cuda.synchronize()
            time.sleep(idle_ms * 1e-3)
            r.append(region("bare")[0])
        print("bare after %.1f ms of idling: median %.1f us" % (idle_ms, np.median(r)))
