#!/usr/bin/env python3
"""Perception boxes -> safe inputs, per tick at 65536 agents x 16 boxes (fp32, 64-agent tiles, ring of distinct
batches larger than the Infinity Cache): (a) ingestion pass to prepared slots, then the filter (two launches, the
slots written and read again); (b) the boxes themselves as slots of the box kinds (one launch).

    python tools/bench_boxes.py [--agents 65536]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import _lib, scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled  # noqa: E402
from sccav_cbf_amd.ingest import bbox_as_obs, ingest_bbox  # noqa: E402


def timed(walk, n_per_walk, reps=300):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        walk()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            walk()
        for _ in range(reps):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n_per_walk) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=65536)
    args = ap.parse_args()
    A, M, RING = args.agents, 16, 10
    dev = torch.device("cuda", 0)
    w = scenarios.config4(A=A, nominal="uniform", seed=11)
    rng = np.random.default_rng(3)
    # boxes at the scene's obstacle positions: extents of cars, the ellipse's heading as yaw, the cone's speed
    bbox = np.zeros((M, 2, A, 4), dtype=np.float32)
    bbox[:, 0, :, 0:2] = w["obs"][:, 0, :, 0:2]
    bbox[:, 0, :, 2] = rng.uniform(1.8, 2.6, (M, A))
    bbox[:, 0, :, 3] = rng.uniform(0.8, 1.1, (M, A))
    bbox[:, 1, :, 0] = rng.uniform(-np.pi, np.pi, (M, A))
    bbox[:, 1, :, 1] = rng.uniform(0.0, 8.0, (M, A))
    kind_prep = torch.from_numpy(np.where(w["obs_type"] == _lib.OBS_ELLIPSE, _lib.OBS_ELLIPSE_PREP,
                                          _lib.OBS_CONE_PREP).astype(np.uint8)).to(dev)
    ring = []
    for b in range(RING):
        sh = 64 * 331 * (b + 1)
        bx = torch.from_numpy(np.roll(bbox, sh, axis=2)).to(dev)
        boxes, kind_box = bbox_as_obs(bx.clone(), kind_prep, buffer=0.5)
        ring.append(dict(state=torch.from_numpy(np.roll(w["state"], sh, axis=0)).float().to(dev),
                         u_ref=torch.from_numpy(np.roll(w["u_ref"], sh, axis=0)).float().to(dev),
                         bbox=bx, boxes=planes_to_tiled(boxes),
                         slots=torch.empty((A // 64, M, 2, 64, 4), dtype=torch.float32, device=dev),
                         u=torch.empty((A, 2), device=dev), st=torch.empty((A,), dtype=torch.int32, device=dev)))
    f = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), streamed_obs=True)

    def two_launches():
        for b in ring:
            ingest_bbox(b["bbox"], kind_prep, 0.5, True, b["slots"])
            f.solve(b["state"], b["u_ref"], b["slots"], kind_prep, u_out=b["u"], status=b["st"])

    def one_launch():
        for b in ring:
            f.solve(b["state"], b["u_ref"], b["boxes"], kind_box, u_out=b["u"], status=b["st"])

    def filter_only():
        for b in ring:
            f.solve(b["state"], b["u_ref"], b["slots"], kind_prep, u_out=b["u"], status=b["st"])

    t2 = timed(two_launches, RING)
    u2 = torch.stack([b["u"] for b in ring]).clone()
    t0 = timed(filter_only, RING)
    t1 = timed(one_launch, RING)
    u1 = torch.stack([b["u"] for b in ring])
    infeasible = float(torch.stack([b["st"] for b in ring]).eq(1).float().mean())
    du = (u1 - u2).abs().amax(dim=2).flatten().double().cpu().numpy()
    print("A=%d M=%d fp32: ingestion pass + filter %.2f us per tick; filter on ready slots %.2f us; boxes as slots "
          "(one launch) %.2f us; infeasible %.2f; difference between the two answers (prepared constants evaluated in "
          "double by the pass, in fp32 by the filter): median %.1e, 99.9 %% %.1e" %
          (A, M, t2, t0, t1, infeasible, np.median(du), np.quantile(du, 0.999)), flush=True)


if __name__ == "__main__":
    main()
