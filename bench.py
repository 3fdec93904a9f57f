#!/usr/bin/env python3
"""Benchmark of the batched CBF-QP safety filter on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one control tick (barrier rows + 2-input QP) for one batch of 65536 agents x
16 obstacle slots (8 ellipse + 8 collision cone, fp32) per GPU.  Inputs are resident in
HBM before the timed region; the steps walk a ring of distinct batches larger than the
256 MiB Infinity Cache so every step streams its inputs from HBM.  Ranks hold independent
batches (weak scaling, no data-path collective).  Rank 0 prints ONE JSON line.
Order: untimed clock ramp (--ramp-ms of work, an idle GPU is slow for its first milliseconds),
W untimed warm-up steps, barrier, exactly K timed steps, barrier.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled  # noqa: E402

AGENTS = 65536
ROWS = 16
RING = 10                     # distinct batches: 10 x 35.9 MB > 256 MiB
HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak


def algorithmic_bytes(A, M, w=4):
    # per QP: state 4 + u_ref 2 + 8 M obstacle scalars + u_out 2, plus the int32 status
    return A * (w * (4 + 2 + 8 * M + 2) + 4)


def make_ring(rank, device, n=RING, agents=AGENTS, tiled=True):
    ring = []
    for b in range(n):
        w = scenarios.config4(A=agents, seed=1234 + 4 + 1000 * rank + b)
        t = dict(state=torch.from_numpy(w["state"]).float().to(device),
                 u_ref=torch.from_numpy(w["u_ref"]).float().to(device),
                 obs=torch.from_numpy(w["obs"]).float().to(device),   # (M,2,A,4) planes
                 obs_type=torch.from_numpy(w["obs_type"]).to(device),
                 u_out=torch.empty((agents, 2), dtype=torch.float32, device=device),
                 status=torch.empty((agents,), dtype=torch.int32, device=device))
        if tiled:   # tiles of 64 agents: (A/64, M, 2, 64, 4)
            t["obs"] = planes_to_tiled(t["obs"])
        ring.append(t)
    return ring


def cpu_baseline(seconds=10.0):
    """The fp64 oracle (OpenMP) on the host cores, on the first batch of the ring."""
    from oracle import oracle as orc
    orc.build()
    threads = min(16, os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    threads = int(os.environ["OMP_NUM_THREADS"])
    w = scenarios.config4(A=AGENTS, seed=1234 + 4)
    p = orc.make_params(alpha=1.0, R=(0.5, 0, 0, 0.5))
    orc.solve_batch(p, w["state"], w["u_ref"], w["obs"], w["obs_type"])  # warm
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.solve_batch(p, w["state"], w["u_ref"], w["obs"], w["obs_type"])
        n += 1
    dt = time.perf_counter() - t0
    out = {"value": n * AGENTS / dt, "unit": "solves/s", "cores": threads, "kind": "port",
           "sample": "%d passes of the fp64 C oracle (OpenMP, %d threads) over one 65536-agent x 16-row "
                     "batch, %.1f s" % (n, threads, dt)}
    # the reference hands every QP to an interpreted general-purpose solver (cvxopt.solvers.cp,
    # cbf/cbf.py:213); cvxopt is not installed here, so scipy's SLSQP on the same rows stands in
    try:
        out["interpreted_solver_proxy"] = slsqp_proxy(orc, p, w)
    except Exception as e:  # scipy missing or failing must not cost the bench line
        out["interpreted_solver_proxy"] = {"error": repr(e)}
    return out


def slsqp_proxy(orc, p, w, n_qp=192, budget=6.0):
    """One scipy SLSQP call per QP, rows from the oracle, single process."""
    from scipy.optimize import minimize
    idx = np.arange(n_qp)
    p = orc.make_params(alpha=1.0, R=(0.5, 0, 0, 0.5), flags=orc.FLAG_SLIP_IO)  # rows and u_ref in QP coordinates
    _, st, rows = orc.solve_batch(p, w["state"][idx], w["u_ref"][idx],
                                  np.ascontiguousarray(w["obs"][:, :, idx, :]), w["obs_type"], want_rows=True)
    R = np.diag([0.5, 0.5])
    done, t0 = 0, time.perf_counter()
    for a in idx:
        un = w["u_ref"][a]
        G, b = rows[:, a - idx[0], :2], rows[:, a - idx[0], 2]
        minimize(lambda u: (u - un) @ R @ (u - un), un, jac=lambda u: 2 * R @ (u - un), method="SLSQP",
                 constraints=[{"type": "ineq", "fun": lambda u: G @ u - b, "jac": lambda u: G}])
        done += 1
        if time.perf_counter() - t0 > budget:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "solves/s", "cores": 1,
            "solver": "scipy.optimize.minimize(method='SLSQP') per QP on the oracle's rows (cvxopt absent)",
            "sample": "%d QPs, %.1f s" % (done, dt)}


def large_batch(flt, device, agents=1 << 20, n=3, reps=60):
    """The same kernel on 2^20 agents per launch (not part of `value`): at that size a launch is
    no longer dominated by its start-up and tail.  The ring re-uses one C4 batch rotated by whole
    tiles: 3 x 575 MB, more than twice the Infinity Cache."""
    w = scenarios.config4(A=agents, seed=99)
    ring = []
    for b in range(n):
        sh = 64 * 1237 * (b + 1)
        ring.append(dict(
            state=torch.from_numpy(np.roll(w["state"], sh, axis=0)).float().to(device),
            u_ref=torch.from_numpy(np.roll(w["u_ref"], sh, axis=0)).float().to(device),
            obs=planes_to_tiled(torch.from_numpy(np.roll(w["obs"], sh, axis=2)).float().to(device)),
            u_out=torch.empty((agents, 2), dtype=torch.float32, device=device),
            status=torch.empty((agents,), dtype=torch.int32, device=device)))
    ty = torch.from_numpy(w["obs_type"]).to(device)
    st = torch.cuda.Stream(device)
    with torch.cuda.stream(st):
        def walk():
            for b in ring:
                flt.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=b["u_out"], status=b["status"])
        walk()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            walk()
        for _ in range(reps):   # the GPU idled while the host built the batch: let the clocks come back
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (reps * n)
    ab = algorithmic_bytes(agents, ROWS)
    return {"agents": agents, "kernel_ms": ms, "solves_per_s": agents / (ms * 1e-3),
            "achieved_GBps": ab / (ms * 1e-3) / 1e9, "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "launches": reps * n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=5000)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per QP (0 = library default)")
    ap.add_argument("--agents", type=int, default=AGENTS)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--planes", action="store_true", help="obstacles as (M,2,A,4) planes instead of 64-agent tiles")
    ap.add_argument("--no-large", action="store_true", help="skip the extra 1M-agent measurement (N=1 only)")
    ap.add_argument("--ramp-ms", type=float, default=60.0,
                    help="untimed work ahead of the warm-up steps that lets an idle GPU reach its clocks")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    A, M = args.agents, ROWS
    flt = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), lanes_per_qp=args.lanes)
    ring = make_ring(rank, device, agents=A, tiled=not args.planes)

    def step(i):
        b = ring[i % RING]
        flt.solve(b["state"], b["u_ref"], b["obs"], b["obs_type"], u_out=b["u_out"], status=b["status"])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # one hipGraph = one walk over the ring; eager launches cover the remainder
    graph = None
    stream = torch.cuda.Stream(device)
    with torch.cuda.stream(stream):
        for i in range(RING):
            step(i)
        torch.cuda.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for i in range(RING):
                    step(i)

        def run(k):
            done = 0
            if graph is not None:
                for _ in range(k // RING):
                    graph.replay()
                done = (k // RING) * RING
            for i in range(done, k):
                step(i)

        # An idle GPU needs some ten milliseconds of work to reach its clocks, more than a short
        # warm-up gives it: untimed walks over the ring first, then the W warm-up steps.
        ramp_steps = 0
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            run(10 * RING)
            torch.cuda.synchronize()
            ramp_steps += 10 * RING
        run(args.warmup)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        run(args.steps)
        ev1.record(stream)
        barrier()
        t1 = time.perf_counter()
        wall = t1 - t0
        dev_ms = ev0.elapsed_time(ev1)

    tw = torch.tensor([wall], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    wall = float(tw.item())

    # Outside the timed region: what collecting every rank's answers costs per step (the solve
    # itself never communicates).  all_gather of u_out (A,2) f32 and status (A,) i32 over RCCL.
    gather_ms = None
    if dist is not None:
        try:
            gu = torch.empty((world * A, 2), dtype=torch.float32, device=device)
            gs = torch.empty((world * A,), dtype=torch.int32, device=device)
            reps = 20
            for i in range(3 + reps):
                if i == 3:
                    torch.cuda.synchronize()
                    tg = time.perf_counter()
                dist.all_gather_into_tensor(gu, ring[0]["u_out"])
                dist.all_gather_into_tensor(gs, ring[0]["status"])
            torch.cuda.synchronize()
            tg = torch.tensor([(time.perf_counter() - tg) / reps * 1e3], dtype=torch.float64, device=device)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            gather_ms = float(tg.item())
        except Exception as e:  # never at the price of the bench line
            print("gather timing skipped: %r" % (e,), file=sys.stderr)

    status = torch.stack([b["status"] for b in ring]).cpu().numpy()
    counts = np.bincount(status.reshape(-1), minlength=3)

    if rank == 0:
        total_solves = A * args.steps * world
        value = total_solves / wall
        kern_ms = dev_ms / args.steps           # device time per step over the timed region
        abytes = algorithmic_bytes(A, M)
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "CBF-QP solves/sec (2-input, 16-constraint)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C4: %d agents x 16 obstacle rows (8 ellipse + 8 collision cone) per GPU, "
                                   "DBM filter, alpha=1, R=diag(.5,.5), ring of %d distinct batches" % (A, RING),
                       "agents_per_gpu": A, "rows": M, "lanes_per_qp": args.lanes,
                       "obs_layout": "planes (M,2,A,4)" if args.planes else "tiles (A/64,M,2,64,4)",
                       "launch": "eager" if graph is None else "hipGraph of %d launches" % RING,
                       "clock_ramp_steps_before_warmup": ramp_steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": abytes},
            "status_counts": {"optimal": int(counts[0]), "infeasible": int(counts[1]),
                              "nonfinite": int(counts[2])},
        }
        if gather_ms is not None:
            out["gather_ms_per_step"] = gather_ms   # not part of value / ms_per_step
        if world == 1 and not args.no_large and A == AGENTS:
            try:
                del ring
                torch.cuda.empty_cache()
                out["roofline_large_batch"] = large_batch(flt, device)
            except Exception as e:  # extra information only
                out["roofline_large_batch"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
