#!/usr/bin/env python3
"""One-process sweep of the filter kernel over batch size x lanes-per-QP x obstacle layout.

    python tools/sweep.py [--sizes 4096,65536,...] [--lanes 1,2,4,8,16] [--dtype f32]

For every point a ring of distinct C4 batches (larger than the Infinity Cache) is captured
in one hipGraph and replayed; prints microseconds per launch and the fraction of 8 TB/s the
algorithmic bytes reach.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,16384,65536,262144,1048576")
    ap.add_argument("--lanes", default="0,1,2,4,8,16")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--reps", type=int, default=0)
    ap.add_argument("--layouts", default="planes,tiles")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    w = 4 if args.dtype == "f32" else 8
    M = args.rows
    for A in [int(s) for s in args.sizes.split(",")]:
        abytes = A * (w * (4 + 2 + 8 * M + 2) + 4)
        n = max(3, min(12, -(-600_000_000 // abytes)))
        base = scenarios.config4(A=A, seed=77)
        ring = []
        g = torch.Generator(device="cpu").manual_seed(A)
        for b in range(n):
            # distinct batches: the same scenario with the agents rotated (cheap to make)
            sh = int(torch.randint(0, A, (1,), generator=g))
            obs = torch.from_numpy(np.roll(base["obs"], sh, axis=2)).to(dt).to(dev)
            if M != 16:
                reps = -(-M // 16)
                obs = obs.repeat(reps, 1, 1, 1)[:M].contiguous()
            t = dict(state=torch.from_numpy(np.roll(base["state"], sh, axis=0)).to(dt).to(dev),
                     u_ref=torch.from_numpy(np.roll(base["u_ref"], sh, axis=0)).to(dt).to(dev),
                     planes=obs, tiles=planes_to_tiled(obs),
                     u_out=torch.empty((A, 2), dtype=dt, device=dev),
                     status=torch.empty((A,), dtype=torch.int32, device=dev))
            ring.append(t)
        ty = torch.from_numpy(np.tile(base["obs_type"], -(-M // 16))[:M].copy()).to(dev)
        reps = args.reps or max(20, min(400, int(4e9 // (abytes * n))))
        for layout in args.layouts.split(","):
            for L in [int(s) for s in args.lanes.split(",")]:
                flt = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), lanes_per_qp=L)
                st = torch.cuda.Stream(dev)
                with torch.cuda.stream(st):
                    def walk():
                        for b in ring:
                            flt.solve(b["state"], b["u_ref"], b[layout], ty, u_out=b["u_out"], status=b["status"])
                    try:
                        walk()
                    except Exception as e:  # variant not available for this M
                        print("A %8d %-6s lanes %2d  -- %s" % (A, layout, L, e), flush=True)
                        continue
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=st):
                        walk()
                    t0 = time.perf_counter()   # the GPU idled while the host built the ring: let the clocks come back
                    while time.perf_counter() - t0 < 0.05:
                        gr.replay()
                        torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    for _ in range(reps):
                        gr.replay()
                    e1.record(st)
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / (reps * n)
                print("A %8d %-6s lanes %2d  %9.2f us  %.3f of 8 TB/s  %.3e solves/s" %
                      (A, layout, L, us, abytes / (us * 1e-6) / 8e12, A / (us * 1e-6)), flush=True)
                del gr


if __name__ == "__main__":
    main()
