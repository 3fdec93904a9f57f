#!/usr/bin/env python3
"""Launch time of the unicycle-input (DUM) and steer-rate (SADBM) filters on the C4 batch
(65536 agents x 16 rows, fp32), beside the bicycle model on the same planes layout."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF  # noqa: E402

dev = torch.device("cuda:0")
w = scenarios.config4(A=65536, nominal="uniform")
t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in w.items()}
t = {k: (v.float().contiguous() if v.dtype.is_floating_point else v) for k, v in t.items()}
A = 65536
u = torch.empty((A, 2), device=dev)
st = torch.empty((A,), dtype=torch.int32, device=dev)
slip = torch.zeros((A, 2), device=dev)
for name in ("dbm", "dum", "sadbm"):
    f = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), dynamics=name)
    kw = dict(slip=slip, dt=0.1) if name == "sadbm" else {}
    run = lambda: f.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], u_out=u, status=st, **kw)
    for _ in range(50):
        run()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-6s %.2f us per launch, %.3g solves/s" % (name, e0.elapsed_time(e1) / 1000 * 1e3, A * 1000 / (e0.elapsed_time(e1) * 1e-3)), flush=True)
