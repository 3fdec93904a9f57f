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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=65536)
    ap.add_argument("--rows", default="16,64,257,1024")
    ap.add_argument("--lanes", default="0", help="lanes_per_qp values to try (0: the library's choice)")
    ap.add_argument("--models", default="dbm,kbm,dum,sadbm")
    ap.add_argument("--layout", default="tiles", choices=("tiles", "agents", "planes"),
                    help="obstacle layout: 64-agent tiles, per-agent lists (A,M,2,4) or planes (M,2,A,4)")
    ap.add_argument("--sorted", action="store_true", help="slots grouped by kind instead of 8 + 8 repeated")
    ap.add_argument("--branches", type=int, default=1,
                    help="parallel branches of the captured walk: the launches are independent batches")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    A = args.agents
    base = scenarios.prepared(scenarios.config4(A=A, nominal="uniform", seed=3))
    torch.manual_seed(1)
    for M in [int(m) for m in args.rows.split(",")]:
        reps = -(-M // 16)
        ring = []
        nring = max(2, -(-600_000_000 // (A * (32 * M + 36))))   # distinct batches: more than twice the Infinity Cache
        for b in range(nring):
            sh = 64 * 331 * (b + 1)
            obs = torch.from_numpy(np.roll(base["obs"], sh, axis=2)).float().to(dev)
            obs = obs.repeat(reps, 1, 1, 1)[:M].contiguous()
            # copy k of an obstacle stands 10 k .. 10 k + 10 m farther from the agent along the line of sight
            # (lane slots stay as they are), so the feasible share of the batch stays that of 16 rows
            ego = torch.from_numpy(np.roll(base["state"], sh, axis=0)[:, 0:2]).float().to(dev)
            is_lane = torch.from_numpy(np.tile(base["obs_type"], reps)[:M] == LANE).to(dev)
            los = obs[:, 0, :, 0:2] - ego[None]
            los = los / los.norm(dim=2, keepdim=True).clamp_min(1e-3)
            far = (torch.arange(M, device=dev) // 16).float()[:, None, None] * 10.0
            far = far + 10.0 * torch.rand((M, 1, 1), device=dev) * (far > 0)
            obs[:, 0, :, 0:2] += torch.where(is_lane[:, None, None], torch.zeros_like(los), los * far)
            if args.sorted:
                order = torch.from_numpy(np.argsort(np.tile(base["obs_type"], reps)[:M], kind="stable")).to(dev)
                obs = obs[order].contiguous()
            ring.append(dict(state=torch.from_numpy(np.roll(base["state"], 64 * 331 * (b + 1), axis=0)).float().to(dev),
                             u_ref=torch.from_numpy(np.roll(base["u_ref"], 64 * 331 * (b + 1), axis=0)).float().to(dev),
                             obs={"tiles": planes_to_tiled, "agents": planes_to_agents,
                                  "planes": lambda o: o}[args.layout](obs)))
            del obs
        ty = np.tile(base["obs_type"], reps)[:M].copy()
        ty = torch.from_numpy(np.sort(ty, kind="stable") if args.sorted else ty).to(dev)
        u = torch.empty((A, 2), device=dev)
        st = torch.empty((A,), dtype=torch.int32, device=dev)
        slip = torch.zeros((A, 2), device=dev)
        abytes = A * (4 * (4 + 2 + 8 * M + 2) + 4)
        for name, lanes in [(n, int(l)) for n in args.models.split(",") for l in args.lanes.split(",")]:
            f = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), dynamics=name, streamed_obs=True, lanes_per_qp=lanes)
            kw = dict(slip=slip, dt=0.1) if name == "sadbm" else {}

            inner = max(1, 10 // nring)   # about 10 launches per captured walk: a replay's own start is a small share

            def walk():
                for b in ring * inner:
                    f.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=u, status=st, **kw)
            try:
                for _ in range(3):
                    walk()
            except Exception as exc:    # a variant that cannot hold this many rows
                print("M %4d %-6s lanes %2d: %s" % (M, name, lanes, str(exc)[:60]), flush=True)
                continue
            torch.cuda.synchronize()
            from sccav_cbf_amd.batched import capture_independent
            outs = [(torch.empty_like(u), torch.empty_like(st)) for _ in ring * inner]   # launches may overlap
            steps = [(lambda b=b, o=o: f.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=o[0], status=o[1], **kw))
                     for b, o in zip(ring * inner, outs)]
            cs = torch.cuda.Stream(dev)
            with torch.cuda.stream(cs):
                g = capture_independent(steps, args.branches)
            n = max(3, min(100, int(4e9 // abytes)))
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
            print("M %4d %-6s lanes %2d %9.2f us per launch  %.3f of 8 TB/s  infeasible %.2f" %
                  (M, name, lanes, us, abytes / (us * 1e-6) / 8e12, float((outs[0][1] == 1).float().mean())), flush=True)
        del ring
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
