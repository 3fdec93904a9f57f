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

--gpus N with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU, starts N
fresh rank processes (one per GPU, RCCL through torch.distributed "nccl") and relays rank 0's
line; it exits non-zero if fewer than N devices are visible or any rank fails.  Under a launcher
that sets WORLD_SIZE (torch.distributed.run) the process is a rank and --gpus must equal WORLD_SIZE.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sccav_cbf_amd import scenarios  # noqa: E402

AGENTS = 65536
ROWS = 16
RING = 10                     # distinct batches: 10 x 35.9 MB > 256 MiB
HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak


def algorithmic_bytes(A, M, w=4):
    # per QP: state 4 + u_ref 2 + 8 M obstacle scalars + u_out 2, plus the int32 status
    return A * (w * (4 + 2 + 8 * M + 2) + 4)


def c4_batch(agents, seed, slots):
    """One C4 batch: 8 ellipse + 8 collision-cone slots per agent, as the reference's objects hold them
    ("written") or with the per-obstacle constants evaluated ("prepared", kinds 4 and 5: same obstacles,
    same 8 scalars per slot, same rows)."""
    w = scenarios.config4(A=agents, seed=seed)
    return scenarios.prepared(w) if slots == "prepared" else w


def make_ring(rank, device, n, agents=AGENTS, layout="tiles", slots="prepared"):
    from sccav_cbf_amd.batched import planes_to_tiled
    ring = []
    for b in range(n):
        w = c4_batch(agents, 1234 + 4 + 1000 * rank + b, slots)
        t = dict(state=torch.from_numpy(w["state"]).float().to(device),
                 u_ref=torch.from_numpy(w["u_ref"]).float().to(device),
                 obs=torch.from_numpy(w["obs"]).float().to(device),   # (M,2,A,4) planes
                 obs_type=torch.from_numpy(w["obs_type"]).to(device),
                 u_out=torch.empty((agents, 2), dtype=torch.float32, device=device),
                 status=torch.empty((agents,), dtype=torch.int32, device=device))
        if layout == "tiles":   # tiles of 64 agents: (A/64, M, 2, 64, 4)
            t["obs"] = planes_to_tiled(t["obs"])
        ring.append(t)
    return ring


def cpu_baseline(seconds=10.0):
    """The fp64 oracle on the host cores, on the first batch of the ring: OpenMP on the threads this
    process may use (OMP_NUM_THREADS where the box sets it, else every core), and one thread."""
    from oracle import oracle as orc
    orc.build()
    host_cores = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", host_cores))
    w = scenarios.config4(A=AGENTS, seed=1234 + 4)
    p = orc.make_params(alpha=1.0, R=(0.5, 0, 0, 0.5))
    omp = orc.lib()   # liboracle.so links libgomp: its symbols resolve through the same handle

    def timed(nthreads, budget, agents):
        omp.omp_set_num_threads(int(nthreads))
        st, ur = w["state"][:agents], w["u_ref"][:agents]
        ob = np.ascontiguousarray(w["obs"][:, :, :agents, :])
        orc.solve_batch(p, st, ur, ob, w["obs_type"])  # warm
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            orc.solve_batch(p, st, ur, ob, w["obs_type"])
            n += 1
        dt = time.perf_counter() - t0
        return n * agents / dt, n, dt

    v, n, dt = timed(threads, seconds, AGENTS)
    v1, n1, dt1 = timed(1, 0.4 * seconds, AGENTS // 8)
    omp.omp_set_num_threads(threads)
    out = {"value": v, "unit": "solves/s", "cores": threads, "threads": threads, "host_cores": host_cores,
           "value_1thread": v1, "kind": "port",
           "sample": "%d passes of the fp64 C oracle (OpenMP, %d threads of %d host cores) over one "
                     "65536-agent x 16-row batch, %.1f s; one thread: %d passes over 8192 agents, %.1f s"
                     % (n, threads, host_cores, dt, n1, dt1)}
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


def _profile_json(name):
    """A number kept from a counter run of this same workload (profiles/*.json), or None."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path)).get("hbm_bytes_per_launch")
    except Exception:
        return None


def _time_graph(walk, stream, n_per_walk, reps, graph=True, branches=1):
    """ms per launch of walk() (n_per_walk launches) replayed from a hipGraph on `stream` (or launched
    eagerly, graph=False: a profiler then sees every launch).  walk may be a list of n_per_walk independent
    launches instead: with branches > 1 they are dealt to that many parallel branches of the graph."""
    steps = walk if isinstance(walk, (list, tuple)) else None
    if steps is not None:
        def walk():
            for f in steps:
                f()
    walk()
    torch.cuda.synchronize()

    class _Eager:
        replay = staticmethod(walk)
    g = _Eager
    if graph:
        from sccav_cbf_amd.batched import capture_independent
        g = capture_independent(steps if steps is not None else [walk], branches if steps is not None else 1, stream)
    for _ in range(reps):   # the GPU idled while the host built the batch: let the clocks come back
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        g.replay()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n_per_walk)   # ms per launch


def large_batch(flt, device, layout, slots, agents=1 << 20, n=3, reps=200, graph=True):
    """The same kernel on 2^20 agents per launch (not part of `value`): at that size a launch is
    no longer dominated by its start-up and tail.  The ring re-uses one C4 batch rotated by whole
    tiles: 3 x 575 MB, more than twice the Infinity Cache."""
    from sccav_cbf_amd.batched import planes_to_tiled
    w = c4_batch(agents, 99, slots)
    ring = []
    for b in range(n):
        sh = 64 * 1237 * (b + 1)
        obs = torch.from_numpy(np.roll(w["obs"], sh, axis=2)).float().to(device)
        ring.append(dict(
            state=torch.from_numpy(np.roll(w["state"], sh, axis=0)).float().to(device),
            u_ref=torch.from_numpy(np.roll(w["u_ref"], sh, axis=0)).float().to(device),
            obs=planes_to_tiled(obs) if layout == "tiles" else obs,
            u_out=torch.empty((agents, 2), dtype=torch.float32, device=device),
            status=torch.empty((agents,), dtype=torch.int32, device=device)))
    ty = torch.from_numpy(w["obs_type"]).to(device)
    st = torch.cuda.Stream(device)
    with torch.cuda.stream(st):
        def launch(b):
            return lambda: flt.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=b["u_out"], status=b["status"])
        steps = [launch(b) for b in ring]
        ms = _time_graph(steps, st, n, reps, graph=graph)
    ab = algorithmic_bytes(agents, ROWS)
    out = {"agents": agents, "kernel_ms": ms, "solves_per_s": agents / (ms * 1e-3),
           "achieved_GBps": ab / (ms * 1e-3) / 1e9, "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "algorithmic_bytes_per_launch": ab, "traffic": _profile_json("traffic_large.json"),
           "launches": reps * n,
           # (three overlapping launches of this size measured slower: 95.7 us per launch)
           "launch": "one at a time (serial hipGraph)" if graph else "eager"}
    return out


def fp64_lines(device):
    """The reference computes in fp64 throughout: the same filter in fp64 on the two fp64 shapes the
    configurations name, priced like the fp32 line (68 + 64 M bytes per QP).
    c5: 16384 agents x (2 lanes + 6 cones), ring of 30 rotations of one batch (> 256 MiB);
    c2: 1 ego x 32 radial obstacles, one wavefront for the QP: a launch-latency case."""
    from sccav_cbf_amd.batched import BatchedCBF, planes_to_tiled
    out = {}
    flt = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]))
    st = torch.cuda.Stream(device)
    with torch.cuda.stream(st):
        A, n = 16384, 30
        w = scenarios.lanes_and_cones(A=A)
        M = w["obs"].shape[0]
        ty = torch.from_numpy(w["obs_type"]).to(device)
        ring = []
        for b in range(n):
            sh = 64 * 37 * (b + 1)
            ring.append(dict(
                state=torch.from_numpy(np.roll(w["state"], sh, axis=0)).to(device),
                u_ref=torch.from_numpy(np.roll(w["u_ref"], sh, axis=0)).to(device),
                obs=planes_to_tiled(torch.from_numpy(np.roll(w["obs"], sh, axis=2)).to(device)),
                u_out=torch.empty((A, 2), dtype=torch.float64, device=device),
                status=torch.empty((A,), dtype=torch.int32, device=device)))

        def launch(b):
            return lambda: flt.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=b["u_out"], status=b["status"])
        steps = [launch(b) for b in ring]
        ms1 = _time_graph(steps, st, n, 40)
        ms = _time_graph(steps, st, n, 40, branches=4)
        ab = algorithmic_bytes(A, M, w=8)
        out["c5"] = {"workload": "C5 fp64: %d agents x %d rows (2 lanes + 6 cones)" % (A, M), "kernel_ms": ms,
                     "launch": "hipGraph of %d launches in 4 parallel branches (independent batches)" % n,
                     "solves_per_s": A / (ms * 1e-3), "algorithmic_bytes_per_launch": ab,
                     "achieved_GBps": ab / (ms * 1e-3) / 1e9, "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "one_at_a_time": {"kernel_ms": ms1, "frac": ab / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBPS}}
        del ring
        # the bench workload itself in the reference's precision: C4 fp64, ring of 5 batches (358 MB)
        A4, n4 = AGENTS, 5
        w4 = scenarios.prepared(scenarios.config4(A=A4, nominal="uniform", seed=1234 + 4))
        ty4 = torch.from_numpy(w4["obs_type"]).to(device)
        flt4 = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), streamed_obs=True)
        ring4 = []
        for b in range(n4):
            sh = 64 * 131 * (b + 1)
            ring4.append(dict(
                state=torch.from_numpy(np.roll(w4["state"], sh, axis=0)).to(device),
                u_ref=torch.from_numpy(np.roll(w4["u_ref"], sh, axis=0)).to(device),
                obs=planes_to_tiled(torch.from_numpy(np.roll(w4["obs"], sh, axis=2)).to(device)),
                u_out=torch.empty((A4, 2), dtype=torch.float64, device=device),
                status=torch.empty((A4,), dtype=torch.int32, device=device)))

        def launch4(b):
            return lambda: flt4.solve(b["state"], b["u_ref"], b["obs"], ty4, u_out=b["u_out"], status=b["status"])
        steps4 = [launch4(b) for b in ring4] * 2
        ms1 = _time_graph(steps4, st, 2 * n4, 40)
        ms = _time_graph(steps4, st, 2 * n4, 40, branches=3)
        ab = algorithmic_bytes(A4, ROWS, w=8)
        out["c4"] = {"workload": "C4 fp64: %d agents x %d rows (the bench workload in double precision)" % (A4, ROWS),
                     "kernel_ms": ms, "launch": "hipGraph of %d launches in 3 parallel branches" % (2 * n4),
                     "solves_per_s": A4 / (ms * 1e-3), "algorithmic_bytes_per_launch": ab,
                     "achieved_GBps": ab / (ms * 1e-3) / 1e9, "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "one_at_a_time": {"kernel_ms": ms1, "frac": ab / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBPS}}
        del ring4
        w = scenarios.radial_seekers(M=32)
        t = {k: torch.from_numpy(v).to(device) for k, v in w.items()}
        uo = torch.empty((1, 2), dtype=torch.float64, device=device)
        so = torch.empty((1,), dtype=torch.int32, device=device)

        def walk2():
            for _ in range(10):
                flt.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], u_out=uo, status=so)
        ms = _time_graph(walk2, st, 10, 200)
        out["c2"] = {"workload": "C2 fp64: 1 ego x 32 radial rows", "kernel_ms": ms,
                     "note": "launch latency, not bandwidth: one QP per launch"}
    return out


# ------------------------------------------------------------------------------ launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher(args, argv):
    """Start args.gpus rank processes of this script and relay rank 0's JSON line.  This process has
    made no GPU call (counting devices does not initialise the runtime) and makes none."""
    n = args.gpus
    if not args.cpu_stub:
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d asked for, %d GPU(s) visible: not running a smaller job under that "
                  "name" % (n, have), file=sys.stderr)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others waiting in a collective: end them with it
    deadline = time.time() + args.launch_timeout
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed or time.time() > deadline:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.wait()
            print("bench.py: rank exit codes %s%s" % (codes, "" if failed else " (timed out)"), file=sys.stderr)
            return 1
        if any(c is None for c in codes):
            time.sleep(0.05)
    line = procs[0].stdout.read().decode()
    last = [ln for ln in line.splitlines() if ln.strip()]
    if not last:
        print("bench.py: rank 0 printed nothing", file=sys.stderr)
        return 1
    out = json.loads(last[-1])
    if out.get("n_gpus") != n:
        print("bench.py: rank 0 reports n_gpus=%r, expected %d" % (out.get("n_gpus"), n), file=sys.stderr)
        return 1
    print(last[-1], flush=True)
    return 0


# -------------------------------------------------------------------------------- a rank
def worker(args):
    global RING
    RING = args.ring
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or leave WORLD_SIZE unset and "
              "let --gpus start them)" % (args.gpus, world), file=sys.stderr)
        return 2
    stub = args.cpu_stub
    dist = None
    _saved_stdout = None
    if stub:
        device = torch.device("cpu")
    else:
        if torch.cuda.device_count() <= local:
            print("bench.py: rank %d needs GPU %d, %d visible" % (rank, local, torch.cuda.device_count()),
                  file=sys.stderr)
            return 2
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
    # started as a rank (by the launcher below or by torch.distributed.run): the process group is set up even
    # for a single rank, so that the collective path of an N = 1 run under a launcher is the N > 1 path
    if world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # the collective library prints a version banner on stdout when it starts: until the JSON line is
        # due, whatever is written to file descriptor 1 goes to stderr
        sys.stdout.flush()
        _saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        if dist.get_world_size() != world:
            print("bench.py: process group has %d ranks, expected %d" % (dist.get_world_size(), world),
                  file=sys.stderr)
            return 2

    A, M = args.agents, ROWS
    if args.large_only:
        # the 2^20-agent measurement on its own (what tools/prof_large.sh runs under rocprofv3)
        from sccav_cbf_amd.batched import BatchedCBF
        flt = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), lanes_per_qp=args.lanes,
                         streamed_obs=not args.no_stream_hint)
        out = large_batch(flt, device, args.layout, args.slots, reps=args.steps if args.steps < 1000 else 200,
                          graph=not args.no_graph)
        out["large_only"] = True
        print(json.dumps(out), flush=True)
        return 0
    if stub:
        # launcher / collective plumbing rehearsal on the CPU: the "solve" is a copy, nothing is measured
        ring = [dict(u_ref=torch.zeros((A, 2)), u_out=torch.empty((A, 2)),
                     status=torch.zeros((A,), dtype=torch.int32)) for _ in range(RING)]

        def step(i):
            b = ring[i % RING]
            b["u_out"].copy_(b["u_ref"])
        sync = (lambda: None)
        flt = None
    else:
        from sccav_cbf_amd.batched import BatchedCBF
        # every step reads a batch that is not touched again for RING - 1 steps: say so
        flt = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), lanes_per_qp=args.lanes,
                         streamed_obs=not args.no_stream_hint)
        ring = make_ring(rank, device, RING, agents=A, layout=args.layout, slots=args.slots)

        def step(i):
            b = ring[i % RING]
            flt.solve(b["state"], b["u_ref"], b["obs"], b["obs_type"], u_out=b["u_out"], status=b["status"])
        sync = torch.cuda.synchronize

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    # one hipGraph = one walk over the ring; eager launches cover the remainder
    graph = None
    ramp_steps = 0
    if stub:
        for i in range(args.warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        sync()
        wall = time.perf_counter() - t0
        barrier()
        wall_barrier = time.perf_counter() - t0
        dev_ms = wall * 1e3
    else:
        stream = torch.cuda.Stream(device)
        with torch.cuda.stream(stream):
            # launches per captured walk: whole walks over the ring, one replay for a short timed region.
            # The steps are independent (every step has its own batch and its own outputs), so the captured
            # walk deals them to `branches` parallel branches of the graph (side streams forked from and joined
            # to the capture stream): while one launch waits for its data another one computes.
            glen = RING * max(1, -(-min(max(args.steps, 10), 100) // RING))
            nb = 1 if args.no_graph else max(1, args.branches)
            for i in range(glen):
                step(i)
            torch.cuda.synchronize()

            def capture(branches, n=None):
                from sccav_cbf_amd.batched import capture_independent
                return capture_independent([(lambda i=i: step(i)) for i in range(n or glen)], branches, stream)
            if not args.no_graph:
                graph = capture(nb)

            def run(k):
                done = 0
                if graph is not None:
                    for _ in range(k // glen):
                        graph.replay()
                    done = (k // glen) * glen
                for i in range(done, k):
                    step(i)

            # An idle GPU needs some ten milliseconds of work to reach its clocks, more than a short
            # warm-up gives it: untimed walks over the ring first, then the W warm-up steps.
            t_ramp = time.perf_counter()
            while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
                run(10 * glen)
                torch.cuda.synchronize()
                ramp_steps += 10 * glen
            run(args.warmup)
            barrier()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(stream)
            run(args.steps)
            ev1.record(stream)
            # every rank stops its own clock when its own K steps are done (the maximum over the ranks is the
            # job's time); the closing barrier follows, and the time up to its end is reported beside it
            sync()
            wall = time.perf_counter() - t0
            barrier()
            wall_barrier = time.perf_counter() - t0
            dev_ms = ev0.elapsed_time(ev1)
            # the same steps one at a time (a serial graph), outside the timed region: what a caller whose
            # launches depend on each other gets, and what a per-kernel profile shows; and the proof that the
            # overlapped walk wrote the same answers
            serial_ms = None
            if graph is not None and nb > 1:
                got = [(b["u_out"].clone(), b["status"].clone()) for b in ring]
                g1 = capture(1)
                n1 = max(2, min(20, 2000 // glen))
                for _ in range(n1):
                    g1.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(n1):
                    g1.replay()
                e1.record(stream)
                torch.cuda.synchronize()
                serial_ms = e0.elapsed_time(e1) / (n1 * glen)
                same_answers = all(torch.equal(u, b["u_out"]) and torch.equal(st, b["status"])
                                   for (u, st), b in zip(got, ring))
            # a timed region of a few hundred steps is mostly the start of one graph and the closing
            # synchronize: the same measurement over 20000 steps, outside the timed region, beside it
            steady = None
            if graph is not None and args.steps < 5000:
                g100 = capture(nb, 10 * RING)
                for _ in range(50):
                    g100.replay()
                torch.cuda.synchronize()
                n_st = 200
                ts = time.perf_counter()
                for _ in range(n_st):
                    g100.replay()
                torch.cuda.synchronize()
                st_ms = (time.perf_counter() - ts) * 1e3 / (n_st * 10 * RING)
                steady = {"steps": n_st * 10 * RING, "ms_per_step": st_ms, "value": A / (st_ms * 1e-3),
                          "launch": "hipGraphs of %d launches in %d parallel branch(es)" % (10 * RING, nb),
                          "note": "this rank, wall clock, outside the timed region"}
                if nb > 1:
                    # the other way to keep three launches in flight: one SERIAL graph per stream, three streams side
                    # by side (no fork / join inside a graph; a third of the host time per launch)
                    from sccav_cbf_amd.batched import capture_independent
                    ss = [torch.cuda.Stream(device) for _ in range(3)]
                    gs = []
                    for j, s_ in enumerate(ss):
                        s_.wait_stream(stream)
                        with torch.cuda.stream(s_):
                            gs.append(capture_independent([(lambda i=i: step(i)) for i in range(j, 30 * RING, 3)], 1, s_))

                    def walk(n):
                        for _ in range(n):
                            for s_, g_ in zip(ss, gs):
                                with torch.cuda.stream(s_):
                                    g_.replay()
                    walk(20)
                    torch.cuda.synchronize()
                    n_w = 70
                    ts = time.perf_counter()
                    walk(n_w)
                    torch.cuda.synchronize()
                    w_ms = (time.perf_counter() - ts) * 1e3 / (n_w * 30 * RING)
                    steady["three_streams"] = {"steps": n_w * 30 * RING, "ms_per_step": w_ms, "value": A / (w_ms * 1e-3),
                                               "launch": "a serial hipGraph of %d launches on each of 3 streams" % (10 * RING)}
                    for s_ in ss:
                        stream.wait_stream(s_)

    # max over ranks of the timed region; every rank's own time; proof that the collective
    # library saw every rank (a sum of ones over the process group)
    rank_ms = [wall / args.steps * 1e3]
    coll_ranks = 1
    if dist is not None:
        tw = torch.tensor([wall, wall_barrier], dtype=torch.float64, device=device)
        allw = torch.empty((2 * world,), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(allw, tw)
        allw, allb = allw.view(world, 2)[:, 0].contiguous(), allw.view(world, 2)[:, 1]
        wall_barrier = float(allb.max().item())
        one = torch.ones((1,), dtype=torch.int32, device=device)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        sync()
        rank_ms = [float(x) / args.steps * 1e3 for x in allw.cpu()]
        wall = float(allw.max().item())
        coll_ranks = int(one.item())

    # Outside the timed region: what collecting every rank's answers costs per step (the solve
    # itself never communicates).  all_gather of u_out (A,2) f32 and status (A,) i32.
    gather_ms = None
    if dist is not None:
        gu = torch.empty((world * A, 2), dtype=torch.float32, device=device)
        gs = torch.empty((world * A,), dtype=torch.int32, device=device)
        reps = 20
        for i in range(3 + reps):
            if i == 3:
                sync()
                tg = time.perf_counter()
            dist.all_gather_into_tensor(gu, ring[0]["u_out"])
            dist.all_gather_into_tensor(gs, ring[0]["status"])
        sync()
        tg = torch.tensor([(time.perf_counter() - tg) / reps * 1e3], dtype=torch.float64, device=device)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_ms = float(tg.item())

    status = torch.stack([b["status"] for b in ring]).cpu().numpy()
    counts = np.bincount(status.reshape(-1), minlength=3)

    rc = 0
    if rank == 0:
        total_solves = A * args.steps * world
        value = total_solves / wall
        kern_ms = dev_ms / args.steps           # device time per step over the timed region
        abytes = algorithmic_bytes(A, M)
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        layout_name = {"tiles": "tiles (A/64,M,2,64,4)", "planes": "planes (M,2,A,4)"}[args.layout]
        out = {
            "metric": "CBF-QP solves/sec (2-input, 16-constraint)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C4: %d agents x 16 obstacle rows (8 ellipse + 8 collision cone) per GPU, "
                                   "DBM filter, alpha=1, R=diag(.5,.5), ring of %d distinct batches" % (A, RING),
                       "agents_per_gpu": A, "rows": M, "lanes_per_qp": args.lanes,
                       "obs_layout": layout_name,
                       "obs_slots": "prepared (kinds 4, 5: 1/a^2, 1/b^2, cos, sin and the cone's obstacle velocity "
                                    "stored in the slot)" if args.slots == "prepared" else "as written (kinds 0, 1)",
                       "launch": "eager" if graph is None else ("hipGraph of %d launches" % glen if nb == 1 else
                                                                "hipGraph of %d launches in %d parallel branches (the steps "
                                                                "are independent batches: their launches overlap)" % (glen, nb)),
                       "clock_ramp_steps_before_warmup": ramp_steps},
            # achieved = algorithmic bytes of the timed region's launches / its device time (HIP events); with
            # overlapping launches kernel_ms is the time per launch of the walk, not one kernel's own duration
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": _profile_json("traffic.json"),
                         "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": abytes},
            "status_counts": {"optimal": int(counts[0]), "infeasible": int(counts[1]),
                              "nonfinite": int(counts[2])},
            # QPs whose rows admit no input: the filter reports status 1 and hands back the nominal input
            "infeasible_fraction": float(counts[1]) / max(1, int(counts.sum())),
        }
        if not stub and serial_ms is not None:
            out["roofline"]["concurrent_launches"] = nb
            out["roofline"]["note"] = ("independent steps overlap (%d launches in flight): achieved = bytes of the timed "
                                       "region's launches / its device time; kernel_ms is that time per launch, not one "
                                       "kernel's own duration, which one_at_a_time gives (rocprofv3 --kernel-trace "
                                       "serialises concurrent kernels and shows that one)" % nb)
            out["roofline"]["one_at_a_time"] = {
                "kernel_ms": serial_ms, "achieved": abytes / (serial_ms * 1e-3) / 1e9,
                "frac": abytes / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "note": "the same launches from a serial graph: one kernel's own duration, what rocprofv3 "
                        "--kernel-trace shows per kernel and what a chain of dependent launches gets"}
            out["roofline"]["overlapped_answers_equal_serial"] = bool(same_answers)
        if not stub and steady is not None:
            steady["roofline_frac"] = abytes / (steady["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            if "three_streams" in steady:
                steady["three_streams"]["roofline_frac"] = (abytes / (steady["three_streams"]["ms_per_step"] * 1e-3) / 1e9
                                                            / HBM_PEAK_GBPS)
            out["steady_state"] = steady
        if stub:
            out["data"] = "none (cpu stub: launcher rehearsal, nothing measured)"
            out["stub"] = True
            out["roofline"] = None
        if dist is not None:
            out["rccl_ranks" if not stub else "gloo_ranks"] = coll_ranks
            out["ms_per_step_by_rank"] = rank_ms
            # the clock of ms_per_step stops when the rank's own steps are done; this one after the closing barrier
            out["ms_per_step_to_end_of_closing_barrier"] = wall_barrier / args.steps * 1e3
            out["gather_ms_per_step"] = gather_ms   # not part of value / ms_per_step
            if coll_ranks != world:
                print("bench.py: the collective saw %d ranks, expected %d" % (coll_ranks, world), file=sys.stderr)
                rc = 1
        if world == 1 and not stub and not args.no_large and A == AGENTS:
            try:
                del ring
                torch.cuda.empty_cache()
                out["roofline_large_batch"] = large_batch(flt, device, args.layout, args.slots)
            except Exception as e:  # extra information only
                out["roofline_large_batch"] = {"error": repr(e)}
        if world == 1 and not stub and not args.no_fp64 and A == AGENTS:
            try:
                out["fp64"] = fp64_lines(device)
            except Exception as e:
                out["fp64"] = {"error": repr(e)}
        if world == 1 and not stub and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        if _saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(_saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=5000)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per QP (0 = library default)")
    ap.add_argument("--agents", type=int, default=AGENTS)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="length of the all-threads sample of the CPU baseline (the one-thread sample takes 0.4 of it)")
    ap.add_argument("--planes", action="store_true", help="obstacles as (M,2,A,4) planes instead of 64-agent tiles")
    ap.add_argument("--slots", choices=("prepared", "written"), default="prepared",
                    help="ellipse / cone slots with their per-obstacle constants evaluated (default) or as the "
                         "reference's objects hold them; 8 scalars per slot either way")
    ap.add_argument("--no-stream-hint", action="store_true",
                    help="do not tell the library that each step's obstacle slots are read once (SCCBF_FLAG_OBS_STREAMED)")
    ap.add_argument("--branches", type=int, default=3,
                    help="parallel branches of the captured walk (independent steps overlap); 1 = one launch at a time")
    ap.add_argument("--ring", type=int, default=RING,
                    help="distinct batches the steps walk over (default 10: more than the Infinity Cache holds, every "
                         "step streams from HBM; 1: the same batch every step, served from the Infinity Cache, which "
                         "shows what the arithmetic alone costs)")
    ap.add_argument("--no-large", action="store_true", help="skip the extra 1M-agent measurement (N=1 only)")
    ap.add_argument("--large-only", action="store_true",
                    help="only the 2^20-agent measurement (with --steps < 1000: that many walks over its ring of 3)")
    ap.add_argument("--no-fp64", action="store_true", help="skip the fp64 lines (N=1 only)")
    ap.add_argument("--ramp-ms", type=float, default=60.0,
                    help="untimed work ahead of the warm-up steps that lets an idle GPU reach its clocks")
    ap.add_argument("--cpu-stub", action="store_true",
                    help="rehearse launcher and collectives on the CPU (gloo, the solve is a copy): measures nothing")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="seconds the --gpus launcher waits for its ranks")
    args = ap.parse_args()
    args.layout = "planes" if args.planes else "tiles"
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launcher(args, sys.argv[1:])
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
