#!/usr/bin/env python3
"""Where a one-launch rollout and the same ticks launched one by one part ways (fp64): per tick, the largest
difference of the worst agent's inputs.  A difference that starts at the last bit and grows by a constant factor per
tick is the closed loop's own sensitivity; one that appears at full size on some tick is a defect."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_nominal as tn  # noqa: E402
from sccav_cbf_amd.nominal import vehicle_step  # noqa: E402

M, motion, tiled, dtype = 33, "constant_velocity", False, torch.float64
A, T = 300, 25
make = tn._rollout_case(A, M, 50 + M, dtype, tiled)
la, obs_a, ty = make()
tr = []
for _ in range(T):
    la.nominal.control(la.state, la.traj, la.target_idx, la.pid_state, u_ref=la.u_ref)
    u, st = la.cbf.solve(la.state, la.u_ref, la.obs, ty, u_out=la.u, status=la.status)
    vehicle_step(la.state, la.u, lr=1.45, lf=1.45, max_steer=la.max_steer, dt=la.dt, beta_out=la.beta)
    tr.append(torch.cat([u, la.beta[:, None], st.to(dtype)[:, None]], dim=1).clone())
    tn._advance_planes(obs_a, ty, la.dt)
    la.obs = obs_a
tr = torch.stack(tr)
lb, _, _ = make()
out = lb.rollout(T, motion=motion, trace=True)
torch.cuda.synchronize()
scale = 1 + tr[:, :, :2].abs().amax(dim=(0, 2))
d = (out["trace"][:, :, :3] - tr[:, :, :3]).abs().amax(dim=2) / scale[None, :]
worst = int(d.amax(dim=0).argmax())
print("worst agent", worst, "scale", float(scale[worst]))
for t in range(T):
    print("tick %2d  diff %.3e  status %d / %d   u = %s" % (t, float(d[t, worst]), int(out["trace"][t, worst, 3]), int(tr[t, worst, 3]),
                                                        tr[t, worst, :3].cpu().numpy()))
print("agents above 1e-9:", int((d.amax(dim=0) > 1e-9).sum()), "above 1e-7:", int((d.amax(dim=0) > 1e-7).sum()))
