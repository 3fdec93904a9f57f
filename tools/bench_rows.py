#!/usr/bin/env python3
"""Launch time of the filter over the number of rows per QP, for every model:
65536 agents (fp32; slots in 64-agent tiles, or --layout agents / planes), M = 16, 64, 257, 1024; the library picks the kernel (4 lanes per QP up
to 64 rows, 16 up to 256, one wavefront per QP beyond).

    python tools/bench_rows.py [--agents 65536] [--rows 16,64,257,1024] [--layout agents]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_agents, planes_to_tiled  # noqa: E402
from sccav_cbf_amd._lib import OBS_LANE as LANE  # noqa: E402


def rows_ring(base, M, A, dev, layout="tiles", by_kind=False, nbytes=600_000_000, dtype=torch.float32):
    """Distinct batches of A agents x M slots (more than twice the Infinity Cache together) made from the 16-slot
    batch `base` (scenarios.prepared(config4)): copy k of an obstacle stands 10 k .. 10 k + 10 m farther from the
    agent along the line of sight (lane slots stay as they are), so the feasible share of the batch stays that
    of 16 rows.  Returns (ring of dicts state / u_ref / obs, obs_type)."""
    reps = -(-M // 16)
    ring = []
    w = 4 if dtype == torch.float32 else 8
    nring = max(2, -(-nbytes // (A * w * (8 * M + 9))))
    kinds = np.tile(base["obs_type"], reps)[:M]
    for b in range(nring):
        sh = 64 * 331 * (b + 1)
        obs = torch.from_numpy(np.roll(base["obs"], sh, axis=2)).to(dtype).to(dev)
        obs = obs.repeat(reps, 1, 1, 1)[:M].contiguous()
        ego = torch.from_numpy(np.roll(base["state"], sh, axis=0)[:, 0:2]).to(dtype).to(dev)
        is_lane = torch.from_numpy(kinds == LANE).to(dev)
        los = obs[:, 0, :, 0:2] - ego[None]
        los = los / los.norm(dim=2, keepdim=True).clamp_min(1e-3)
        far = (torch.arange(M, device=dev) // 16).to(dtype)[:, None, None] * 10.0
        far = far + 10.0 * torch.rand((M, 1, 1), device=dev, dtype=dtype) * (far > 0)
        obs[:, 0, :, 0:2] += torch.where(is_lane[:, None, None], torch.zeros_like(los), los * far)
        if by_kind:
            order = torch.from_numpy(np.argsort(kinds, kind="stable")).to(dev)
            obs = obs[order].contiguous()
        ring.append(dict(state=torch.from_numpy(np.roll(base["state"], sh, axis=0)).to(dtype).to(dev),
                         u_ref=torch.from_numpy(np.roll(base["u_ref"], sh, axis=0)).to(dtype).to(dev),
                         obs={"tiles": planes_to_tiled, "agents": planes_to_agents,
                              "planes": lambda o: o}[layout](obs)))
        del obs
    ty = torch.from_numpy(np.sort(kinds, kind="stable") if by_kind else kinds.copy()).to(dev)
    return ring, ty


def time_rows(f, ring, ty, A, M, branches=1, **kw):
    """us per launch of filter f over the ring, one launch at a time from a captured walk (about ten launches per
    replay), and the status of the first batch."""
    from sccav_cbf_amd.batched import capture_independent
    dev = ty.device
    nring = len(ring)
    w = ring[0]["state"].element_size()
    abytes = A * (w * (4 + 2 + 8 * M + 2) + 4)
    inner = max(1, 10 // nring)   # about 10 launches per captured walk: a replay's own start is a small share
    outs = [(torch.empty((A, 2), device=dev, dtype=ring[0]["state"].dtype), torch.empty((A,), dtype=torch.int32, device=dev))
            for _ in ring * inner]   # launches may overlap
    steps = [(lambda b=b, o=o: f.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=o[0], status=o[1], **kw))
             for b, o in zip(ring * inner, outs)]
    for _ in range(3):
        for st in steps:
            st()
    torch.cuda.synchronize()
    cs = torch.cuda.Stream(dev)
    with torch.cuda.stream(cs):
        g = capture_independent(steps, branches)
    n = max(3, min(100, int(4e9 // abytes)))
    import time
    t0 = time.perf_counter()       # the GPU needs tens of milliseconds of work to reach its clocks
    while time.perf_counter() - t0 < 0.08:
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (nring * inner * n)
    return us, abytes / (us * 1e-6) / 8e12, outs[0][1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=65536)
    ap.add_argument("--rows", default="16,64,257,1024")
    ap.add_argument("--lanes", default="0", help="lanes_per_qp values to try (0: the library's choice)")
    ap.add_argument("--models", default="dbm,kbm,dum,sadbm")
    ap.add_argument("--layout", default="tiles", choices=("tiles", "agents", "planes"),
                    help="obstacle layout: 64-agent tiles, per-agent lists (A,M,2,4) or planes (M,2,A,4)")
    ap.add_argument("--dtype", default="f32", choices=("f32", "f64"))
    ap.add_argument("--sorted", action="store_true", help="slots grouped by kind instead of 8 + 8 repeated")
    ap.add_argument("--branches", type=int, default=1,
                    help="parallel branches of the captured walk: the launches are independent batches")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    A = args.agents
    base = scenarios.prepared(scenarios.config4(A=A, nominal="uniform", seed=3))
    torch.manual_seed(1)
    for M in [int(m) for m in args.rows.split(",")]:
        dt = torch.float32 if args.dtype == "f32" else torch.float64
        ring, ty = rows_ring(base, M, A, dev, args.layout, args.sorted, dtype=dt)
        slip = torch.zeros((A, 2), device=dev, dtype=dt)
        for name, lanes in [(n, int(l)) for n in args.models.split(",") for l in args.lanes.split(",")]:
            f = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), dynamics=name, streamed_obs=True, lanes_per_qp=lanes)
            kw = dict(slip=slip, dt=0.1) if name == "sadbm" else {}
            try:
                us, frac, st = time_rows(f, ring, ty, A, M, args.branches, **kw)
            except Exception as exc:    # a variant that cannot hold this many rows
                print("M %4d %-6s lanes %2d: %s" % (M, name, lanes, str(exc)[:60]), flush=True)
                continue
            print("M %4d %-6s lanes %2d %9.2f us per launch  %.3f of 8 TB/s  infeasible %.2f" %
                  (M, name, lanes, us, frac, float((st == 1).float().mean())), flush=True)
        del ring
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
