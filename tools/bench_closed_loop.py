#!/usr/bin/env python3
"""Time the nominal controller, the filter, the vehicle step and a captured closed-loop tick.

    python tools/bench_closed_loop.py [--agents 65536] [--dtype f32]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled  # noqa: E402
from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop, vehicle_step  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=65536)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--far", action="store_true", help="obstacles 1000 m away: no row ever active")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    A = args.agents
    course = np.load(os.path.join(ROOT, "tests", "golden", "sce_course.npz"))   # the shipped 2034-point course
    cx, cy, cyaw = course["cx"], course["cy"], course["cyaw"]
    traj = torch.from_numpy(np.stack([cx, cy, cyaw, np.full_like(cx, 30 / 3.6)], axis=1)).to(dt).to(dev)
    w = scenarios.config4(A=A, seed=1)
    # agents start within a few metres of the course (the closed-loop use case); the obstacles of the
    # C4 batch keep their position relative to their agent
    rng = np.random.default_rng(2)
    at = rng.integers(0, len(cx), A)
    xy = np.stack([cx[at] + rng.normal(0, 3, A), cy[at] + rng.normal(0, 3, A)], axis=1)
    w["obs"][:, 0, :, 0:2] += (xy - w["state"][:, 0:2])[None, :, :]
    w["state"][:, 0:2] = xy
    if args.far:
        w["obs"][:, 0, :, 0] += 1000.0
    w["state"][:, 2] = cyaw[at] + rng.normal(0, 0.3, A)
    state = torch.from_numpy(w["state"]).to(dt).to(dev)
    state0 = state.clone()
    obs = planes_to_tiled(torch.from_numpy(w["obs"]).to(dt).to(dev))
    ty = torch.from_numpy(w["obs_type"]).to(dev)
    cbf = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]))
    nom = BatchedNominal(axle=2.9, k=0.5, ks=0.0)
    loop = ClosedLoop(cbf, nom, traj, state, obs, ty)
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        loop.state.copy_(state0)
        t_nom = timed(lambda: nom.control(loop.state, traj, loop.target_idx, None, u_ref=loop.u_ref), 200)
        t_flt = timed(lambda: cbf.solve(loop.state, loop.u_ref, obs, ty, u_out=loop.u, status=loop.status), 200)
        t_stp = timed(lambda: vehicle_step(loop.state, loop.u, beta_out=loop.beta), 200)
        loop.tick()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            loop.state.copy_(state0)        # every replay starts from the same states
            loop.target_idx.zero_()
            for _ in range(10):
                loop.tick()
        t_tick = timed(g.replay, 50) / 10
        # three independent loops (a sweep over scenarios): each loop's ticks stay in order on one branch of the
        # graph, the loops overlap
        from sccav_cbf_amd.batched import capture_independent
        loops = [ClosedLoop(cbf, nom, traj, state0.clone(), obs, ty) for _ in range(3)]
        for lp in loops:
            lp.tick()
        torch.cuda.synchronize()

        def ten_ticks(lp):
            def run():
                lp.state.copy_(state0)
                lp.target_idx.zero_()
                for _ in range(10):
                    lp.tick()
            return run
        g3 = capture_independent([ten_ticks(lp) for lp in loops], 3)
        t_tick3 = timed(g3.replay, 50) / 30
        # ONE batch cut into three parts on tile boundaries: agents never depend on each other, so the parts are three
        # independent loops whose launches overlap, while every part's own ticks stay in order
        tiles = A // 64
        cuts = [0, (tiles // 3) * 64, (2 * tiles // 3) * 64, A]
        parts = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            st0 = state0[lo:hi].clone()
            lp = ClosedLoop(cbf, nom, traj, st0.clone(), obs[lo // 64:hi // 64].contiguous(), ty)
            lp.tick()
            parts.append((lp, st0))
        torch.cuda.synchronize()

        def ten_ticks_part(lp, st0):
            def run():
                lp.state.copy_(st0)
                lp.target_idx.zero_()
                for _ in range(10):
                    lp.tick()
            return run
        gs = capture_independent([ten_ticks_part(lp, st0) for lp, st0 in parts], 3)
        t_split = timed(gs.replay, 50) / 10
        # the same ticks fused into one launch per 100
        T = 100
        loop.state.copy_(state0)
        loop.target_idx.zero_()
        loop.rollout(T)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            loop.state.copy_(state0)
            loop.target_idx.zero_()
            loop.rollout(T)
        e1.record()
        torch.cuda.synchronize()
        t_roll = e0.elapsed_time(e1) * 1e3 / (reps * T)
        e0.record()
        for _ in range(reps):
            loop.state.copy_(state0)
            loop.target_idx.zero_()
            loop.rollout(T, motion="constant_velocity")
        e1.record()
        torch.cuda.synchronize()
        t_roll_cv = e0.elapsed_time(e1) * 1e3 / (reps * T)
        # one tick per launch through the rollout kernel (nominal + filter + step in one launch, the
        # obstacles re-read every tick as when the host moves them in between)
        loop.state.copy_(state0)
        loop.target_idx.zero_()
        out1 = loop.rollout(1, trace=True)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=st):
            loop.state.copy_(state0)
            loop.target_idx.zero_()
            for _ in range(10):
                loop.rollout(1, trace=True, out=out1)
        t_one = timed(g1.replay, 50) / 10
    print("   one-tick rollout launches, captured: %.2f us per tick" % t_one)
    with torch.cuda.stream(st):
        pass
    print("   fused rollout (%d ticks per launch): %.2f us per tick (%.3e agent-ticks/s); with moving obstacles %.2f us"
          % (T, t_roll, A / (t_roll * 1e-6), t_roll_cv))
    with torch.cuda.stream(st):
        pass
    print("A=%d N=%d %s: nominal %.2f us, filter %.2f us, vehicle step %.2f us, captured tick %.2f us "
          "(%.3e agent-ticks/s); three independent loops overlapping: %.2f us per tick (%.3e agent-ticks/s); this "
          "batch cut into three parts that overlap: %.2f us per tick of the whole batch"
          % (A, traj.shape[0], args.dtype, t_nom, t_flt, t_stp, t_tick, A / (t_tick * 1e-6), t_tick3,
             A / (t_tick3 * 1e-6), t_split))


if __name__ == "__main__":
    main()
