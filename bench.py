#!/usr/bin/env python3
"""Benchmark of the batched CBF-QP safety filter on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one control tick (barrier rows + 2-input QP) for one batch of 65536 agents x 16 obstacle slots
(8 ellipse + 8 collision cone, fp32).  Inputs are resident in HBM before the timed region; the steps walk a
ring of 20 distinct batches (718 MB, more than twice the 256 MiB Infinity Cache), so every step streams its
inputs from HBM.  The steps are independent batches (a Monte-Carlo sweep: every step has its own inputs and
outputs), and the timed region hands them to the library twenty at a time: ONE launch of
sccbf_solve_batches_f32 over twenty batch records (the answers of twenty sccbf_solve_f32 launches, bit for bit).
--scaling weak (default): every rank owns a full ring of 65536-agent batches; strong: ONE 65536-agent batch per
step, every rank solves its contiguous block of 65536 / N agents (the 65536-agent x 16-row configuration as the baseline states it).  Either
way the solve has no collective.  Rank 0 prints ONE JSON line.
Order: untimed clock ramp (--ramp-ms of work, an idle GPU is slow for its first milliseconds), W untimed
warm-up steps, barrier, exactly K timed steps, barrier.

--gpus N with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU, starts N fresh rank
processes (one per GPU, RCCL through torch.distributed "nccl") and relays rank 0's line; it exits non-zero if
fewer than N devices are visible or any rank fails.  Under a launcher that sets WORLD_SIZE
(torch.distributed.run) the process is a rank and --gpus must equal WORLD_SIZE.
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
RING = 20                     # distinct batches = batches per launch: 20 x 35.9 MB = 718 MB
HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak


def algorithmic_bytes(A, M, w=4):
    # per QP: state 4 + u_ref 2 + 8 M obstacle scalars + u_out 2, plus the int32 status
    return A * (w * (4 + 2 + 8 * M + 2) + 4)


def frac_of_peak(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS


def c4_batch(agents, seed, slots):
    """One C4 batch: 8 ellipse + 8 collision-cone slots per agent, as the reference's objects hold them
    ("written") or with the per-obstacle constants evaluated ("prepared", kinds 4 and 5: same obstacles,
    same 8 scalars per slot, same rows)."""
    w = scenarios.config4(A=agents, seed=seed)
    return scenarios.prepared(w) if slots == "prepared" else w


def to_device(w, device, dtype=torch.float32, layout="tiles", lo=0, hi=None):
    """Host batch (or the agents [lo, hi) of it) -> device tensors with outputs."""
    from sccav_cbf_amd.batched import planes_to_tiled
    hi = w["state"].shape[0] if hi is None else hi
    n = hi - lo
    t = dict(state=torch.from_numpy(np.ascontiguousarray(w["state"][lo:hi])).to(device).to(dtype),
             u_ref=torch.from_numpy(np.ascontiguousarray(w["u_ref"][lo:hi])).to(device).to(dtype),
             obs=torch.from_numpy(np.ascontiguousarray(w["obs"][:, :, lo:hi])).to(device).to(dtype),
             obs_type=torch.from_numpy(w["obs_type"]).to(device),
             u_out=torch.empty((n, 2), dtype=dtype, device=device),
             status=torch.empty((n,), dtype=torch.int32, device=device))
    if layout == "tiles":   # tiles of 64 agents: (A/64, M, 2, 64, 4)
        t["obs"] = planes_to_tiled(t["obs"])
    return t


def make_ring(seed0, device, n, agents=AGENTS, layout="tiles", slots="prepared", lo=0, hi=None):
    return [to_device(c4_batch(agents, seed0 + b, slots), device, layout=layout, lo=lo, hi=hi) for b in range(n)]


def rotated_ring(w, device, n, dtype, stride, layout="tiles"):
    """n batches made of one by rotating its agents by whole tiles: distinct addresses, the same answers."""
    ring = []
    for b in range(n):
        sh = 64 * stride * (b + 1)
        r = dict(state=np.roll(w["state"], sh, axis=0), u_ref=np.roll(w["u_ref"], sh, axis=0),
                 obs=np.roll(w["obs"], sh, axis=2), obs_type=w["obs_type"])
        ring.append(to_device(r, device, dtype=dtype, layout=layout))
    return ring


def shard_views(ring, n):
    """The first n agents (whole tiles) of every batch of a tiled ring, as views: no new memory."""
    return [dict(state=b["state"][:n], u_ref=b["u_ref"][:n], obs=b["obs"][:n // 64], obs_type=b["obs_type"],
                 u_out=b["u_out"][:n], status=b["status"][:n]) for b in ring]


# ------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(seconds=10.0, slots="prepared"):
    """The fp64 oracle on the host cores, on the first batch of the ring (the same slot kinds the GPU reads):
    OpenMP on the threads this process may use (OMP_NUM_THREADS where the box sets it, else every core), and
    one thread."""
    from oracle import oracle as orc
    orc.build()
    host_cores = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", host_cores))
    w = c4_batch(AGENTS, 1234 + 4, slots)
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
                     "65536-agent x 16-row batch (%s slots), %.1f s; one thread: %d passes over 8192 agents, %.1f s"
                     % (n, threads, host_cores, slots, dt, n1, dt1)}
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


# ------------------------------------------------------------------------------ timing helpers
def _profile_json(name):
    """HBM bytes per launch kept from a counter run of this same workload (profiles/*.json), or None when there is
    none or when it was taken on other sources than the library that is loaded now."""
    from sccav_cbf_amd import _build
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        return d.get("hbm_bytes_per_launch") if d.get("source_hash") == _build.source_hash() else None
    except Exception:
        return None


def _events_ms(fn, stream, reps, warm=None, warm_ms=25.0):
    """Device milliseconds per call of fn() (HIP events on `stream`), after `warm` untimed calls and at least
    warm_ms of them (the host work between two measurements lets the GPU's clocks fall)."""
    t0, n = time.perf_counter(), 0
    while n < (reps // 4 + 2 if warm is None else warm) or (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _time_graph(steps, stream, reps, branches=1, graph=True):
    """ms per launch of the independent launches `steps` replayed from a hipGraph on `stream` (dealt to that many
    parallel branches), or launched eagerly (graph=False: a profiler then sees every launch)."""
    def walk():
        for f in steps:
            f()
    walk()
    torch.cuda.synchronize()
    fn = walk
    if graph:
        from sccav_cbf_amd.batched import capture_independent
        fn = capture_independent(steps, branches, stream).replay
    return _events_ms(fn, stream, reps, warm=reps) / len(steps)


def _solve_fn(flt, b, ty=None):
    ty = b["obs_type"] if ty is None else ty
    return lambda: flt.solve(b["state"], b["u_ref"], b["obs"], ty, u_out=b["u_out"], status=b["status"])


def _line(A, M, ms, w=4, **kw):
    ab = algorithmic_bytes(A, M, w)
    d = {"us_per_step": ms * 1e3, "solves_per_s": A / (ms * 1e-3), "frac": frac_of_peak(ab, ms)}
    d.update(kw)
    return d


def fused_and_single(flt, ring, stream, reps_fused=60, reps_single=20):
    """The ring's batches as ONE launch (sccbf_solve_batches_*) and one launch at a time (a serial hipGraph of
    sccbf_solve_* launches): ms per batch each."""
    from sccav_cbf_amd.batched import BatchTable
    table = BatchTable(ring)
    fused = _events_ms(lambda: flt.solve_batches(table), stream, reps_fused) / len(ring)
    single = _time_graph([_solve_fn(flt, b) for b in ring], stream, reps_single)
    return fused, single


def _filter(lanes=0, alpha=1.0, streamed=True, **kw):
    from sccav_cbf_amd.batched import BatchedCBF
    return BatchedCBF(alpha=alpha, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), lanes_per_qp=lanes, streamed_obs=streamed, **kw)


# ------------------------------------------------------------------------------ sections beside the headline (N = 1)
def large_batch(flt, device, layout, slots, agents=1 << 20, n=3, reps=200, graph=True):
    """The same kernel on 2^20 agents in ONE sccbf_solve_f32 launch (not part of `value`).  The ring re-uses one C4
    batch rotated by whole tiles: 3 x 575 MB, more than twice the Infinity Cache."""
    ring = rotated_ring(c4_batch(agents, 99, slots), device, n, torch.float32, 1237, layout)
    st = torch.cuda.Stream(device)
    with torch.cuda.stream(st):
        ms = _time_graph([_solve_fn(flt, b) for b in ring], st, reps, graph=graph)
    ab = algorithmic_bytes(agents, ROWS)
    return {"agents": agents, "kernel_ms": ms, "solves_per_s": agents / (ms * 1e-3),
            "achieved_GBps": ab / (ms * 1e-3) / 1e9, "frac": frac_of_peak(ab, ms),
            "algorithmic_bytes_per_launch": ab, "traffic": _profile_json("traffic_large.json"),
            "launches": reps * n, "launch": "one at a time (serial hipGraph)" if graph else "eager"}


def long_lists_section(device):
    """Lists of 128, 257 and 1024 slots per agent at 65536 agents (fp32, DBM), one launch at a time: from per-agent
    lists (SCCBF_FLAG_OBS_AGENT_MAJOR), the layout the kernels for long lists stream fastest from, and from the
    64-agent tiles of the headline."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_rows import rows_ring, time_rows
    from sccav_cbf_amd import scenarios
    base = scenarios.prepared(scenarios.config4(A=AGENTS, nominal="uniform", seed=3))
    flt = _filter()
    out = {}
    for M in (128, 257, 1024):
        row = {}
        for layout in ("agents", "tiles"):
            ring, ty = rows_ring(base, M, AGENTS, device, layout)
            us, frac, st = time_rows(flt, ring, ty, AGENTS, M)
            row[layout] = {"us_per_launch": us, "frac": frac, "solves_per_s": AGENTS / (us * 1e-6),
                           "infeasible_fraction": float((st == 1).float().mean())}
            del ring
            torch.cuda.empty_cache()
        out["rows_%d" % M] = row
    out["note"] = ("65536 agents, fp32, prepared ellipse / cone slots repeated at growing distance; frac = algorithmic bytes "
                   "(32 M + 36 per agent) / launch time / 8 TB/s; agents = obs as (A, M, 2, 4), tiles = (A/64, M, 2, 64, 4)")
    return out


def variants_section(ring, device, stream):
    """Lanes per QP at C4: 64 = the one-wavefront-per-QP variant, 0 = the library's choice (4 at this size)."""
    out = {}
    for lanes in (0, 1, 2, 4, 8, 16, 64):
        fused, single = fused_and_single(_filter(lanes), ring, stream, reps_fused=20 if lanes < 64 else 6,
                                         reps_single=8 if lanes < 64 else 3)
        out["lanes_%d" % lanes if lanes else "auto"] = {
            "twenty_batches_per_launch": _line(AGENTS, ROWS, fused), "one_launch_per_step": _line(AGENTS, ROWS, single)}
    out["note"] = ("lanes per QP (sccbf_params.lanes_per_qp): 1-16 keep the rows in registers, 64 = one wavefront per QP with the "
                   "rows in LDS only; same answers")
    return out


def configs_section(device, c4_f32):
    """Every configuration of the baseline in one object: microseconds per step, solves/s, share of 8 TB/s."""
    from sccav_cbf_amd.batched import BatchedCBF, BatchTable
    out = {}
    st = torch.cuda.Stream(device)
    with torch.cuda.stream(st):
        # C2: 1 ego x 32 radial seekers, fp64: one QP per launch, launch latency
        w = scenarios.radial_seekers(M=32)
        b = to_device(w, device, dtype=torch.float64, layout="planes")
        ms = _time_graph([_solve_fn(BatchedCBF(alpha=1.0, R=np.eye(2)), b)] * 10, st, 200)
        out["c2"] = _line(1, 32, ms, w=8, workload="C2: 1 ego x 32 radial dynamic obstacles, fp64", dtype="f64",
                          launch="one launch per step (serial hipGraph)",
                          note="one QP per launch: launch latency, not bandwidth")
        # C3: 4096 agents x 8 ellipses, fp32, nominal input from the batched Stanley + PID controller
        w = scenarios.config3(A=4096)
        ring = rotated_ring(w, device, 8, torch.float32, 3)
        flt = _filter(streamed=False)
        fused, single = fused_and_single(flt, ring, st, reps_fused=200, reps_single=100)
        out["c3"] = _line(4096, 8, single, workload="C3: 4096 agents x 8 ellipse obstacles, fp32, Stanley nominal",
                          dtype="f32", launch="one launch per step (serial hipGraph)",
                          eight_batches_per_launch=_line(4096, 8, fused))
        try:
            out["c3"]["with_nominal_controller"] = c3_with_nominal(device, st, w, flt)
        except Exception as e:
            out["c3"]["with_nominal_controller"] = {"error": repr(e)}
        # C4 in both precisions
        out["c4"] = {"f32": c4_f32}
        w4 = scenarios.prepared(scenarios.config4(A=AGENTS, nominal="uniform", seed=1234 + 4))
        ring4 = rotated_ring(w4, device, 10, torch.float64, 131)
        fused, single = fused_and_single(_filter(), ring4, st, reps_fused=40, reps_single=20)
        out["c4"]["f64"] = _line(AGENTS, ROWS, fused, w=8, dtype="f64",
                                 workload="C4 fp64: 65536 agents x 16 rows (the bench workload in double precision)",
                                 launch="ten batches per launch (sccbf_solve_batches_f64)",
                                 one_launch_per_step=_line(AGENTS, ROWS, single, w=8))
        del ring4
        # C5: 2 lanes + 6 cones, 16384 agents, fp32 against fp64
        w5 = scenarios.lanes_and_cones(A=16384)
        M5 = w5["obs"].shape[0]
        out["c5"] = {"workload": "C5: 16384 agents x %d rows (2 lanes + 6 collision cones)" % M5}
        for name, dt, wbytes, n in (("f32", torch.float32, 4, 60), ("f64", torch.float64, 8, 30)):
            ring5 = rotated_ring(w5, device, n, dt, 37)
            fused, single = fused_and_single(_filter(), ring5, st, reps_fused=40, reps_single=20)
            out["c5"][name] = _line(16384, M5, fused, w=wbytes, launch="%d batches per launch" % n,
                                    one_launch_per_step=_line(16384, M5, single, w=wbytes))
            del ring5
        sweep = {}
        for alpha in (0.5, 1.0, 5.0):
            res = {}
            for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
                b = to_device(w5, device, dtype=dt, layout="planes")
                rows = BatchedCBF(alpha=alpha, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), slip_io=True).solve(
                    b["state"], b["u_ref"], b["obs"], b["obs_type"], u_out=b["u_out"], status=b["status"], want_rows=True)[2]
                torch.cuda.synchronize()
                res[name] = (b["u_out"].double().cpu().numpy(), b["status"].cpu().numpy(), rows.double().cpu().numpy())
            (u32, s32, _), (u64, s64, r64) = res["f32"], res["f64"]
            both = (s32 == 0) & (s64 == 0)
            du = np.abs(u32 - u64).max(axis=1) / np.maximum(1.0, np.hypot(u64[:, 0], u64[:, 1]))
            # an optimum on two nearly parallel rows moves by (rounding of the rows) / |sin of their angle|: the stated
            # tolerance |du| <= 1e-3 max(1, |u|) holds where |sin| >= 1e-3; the others are compared on cost and feasibility
            well = both & (_active_sin(u64, r64) >= 1e-3)
            sweep["alpha_%g" % alpha] = {"max_rel_du": float(du[well].max()), "p999_rel_du": float(np.quantile(du[well], 0.999)),
                                         "status_agreement": float((s32 == s64).mean()),
                                         "optimal_in_both": int(both.sum()), "well_conditioned": int(well.sum()),
                                         "max_rel_du_all_pairs": float(du[both].max())}
        out["c5"]["fp32_vs_fp64"] = dict(sweep, note="|u32 - u64| / max(1, |u64|), u in QP coordinates (a, beta), over the agents "
                                                     "optimal in both precisions whose active pair of rows has |sin| >= 1e-3")
    return out


def _active_sin(u, rows, tol=1e-6):
    """|sin| of the angle between the two tightest rows at u when both are active (slack below tol), else 1:
    u (A,2), rows (M,A,4) = a0, a1, b, h with a . u >= b."""
    lhs = rows[:, :, 0] * u[None, :, 0] + rows[:, :, 1] * u[None, :, 1]
    nrm = np.hypot(rows[:, :, 0], rows[:, :, 1]) + 1e-300
    slack = (lhs - rows[:, :, 2]) / nrm
    order = np.argsort(slack, axis=0)
    ar = np.arange(u.shape[0])
    i0, i1 = order[0], order[1]
    a0, a1 = rows[i0, ar, :2] / nrm[i0, ar, None], rows[i1, ar, :2] / nrm[i1, ar, None]
    sin = np.abs(a0[:, 0] * a1[:, 1] - a0[:, 1] * a1[:, 0])
    return np.where((slack[i0, ar] < tol) & (slack[i1, ar] < tol), sin, 1.0)


def c3_with_nominal(device, st, w, flt):
    """C3 as a control tick on the device: batched Stanley + PID on the shipped course, then the filter."""
    from sccav_cbf_amd.nominal import BatchedNominal
    cx, cy, cyaw = scenarios.spline_course()
    traj = torch.from_numpy(np.stack([cx, cy, cyaw, np.full_like(cx, scenarios.TARGET_SPEED)], axis=1)).float().to(device)
    nom = BatchedNominal(axle=2.9, k=0.5, ks=0.0, kp=1.0, target_speed=scenarios.TARGET_SPEED)
    b = to_device(w, device)
    A = b["state"].shape[0]
    idx = torch.zeros((A,), dtype=torch.int32, device=device)

    def tick():
        nom.control(b["state"], traj, idx, None, u_ref=b["u_ref"])
        flt.solve(b["state"], b["u_ref"], b["obs"], b["obs_type"], u_out=b["u_out"], status=b["status"])
    ms = _time_graph([tick] * 10, st, 100)
    return {"us_per_step": ms * 1e3, "solves_per_s": A / (ms * 1e-3),
            "launch": "two launches per step (sccbf_nominal_f32, sccbf_solve_f32), serial hipGraph",
            "course_waypoints": int(traj.shape[0])}


def shard_section(flt, ring, stream):
    """One GPU's share of C4 as stated (65536 agents over N GPUs): 8192, 16384 and 32768 agents per step."""
    out = {}
    t_full = None
    for n in (AGENTS, 32768, 16384, 8192):
        fused, single = fused_and_single(flt, shard_views(ring, n), stream, reps_fused=40, reps_single=20)
        if n == AGENTS:
            t_full = (fused, single)
        out["%d_agents" % n] = {
            "gpus_sharing_65536": AGENTS // n,
            "twenty_steps_per_launch": _line(n, ROWS, fused, speedup_vs_one_gpu=t_full[0] / fused),
            "one_launch_per_step": _line(n, ROWS, single, speedup_vs_one_gpu=t_full[1] / single)}
    out["note"] = ("expected strong scaling of C4 from one GPU's own measurements: time per step of a 65536 / N-agent shard "
                   "against the full batch; a launch of 8192 agents is bound by launch latency, twenty such steps in one "
                   "launch are not")
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
        # one build (hipcc only, no GPU call) before the ranks start; they then refuse to build on their own
        from sccav_cbf_amd import _build
        _build.build_library()
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SCCBF_NO_REBUILD="1")
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
class Region:
    """The K steps of one scaling mode on this rank: ring, batch table, and the launches of the timed region."""

    def __init__(self, args, mode, rank, world, device, stub):
        from sccav_cbf_amd.sharded import shard_bounds
        self.mode, self.stub, self.world = mode, stub, world
        A = args.agents
        if mode == "strong":     # one batch per step for the whole job: this rank's contiguous block of whole tiles
            self.lo, self.hi = shard_bounds(A, world, rank, align=64)
            seed0 = 1234 + 4
        else:                    # a full batch per step and rank, made from the rank's own seed
            self.lo, self.hi = 0, A
            seed0 = 1234 + 4 + 1000 * rank
        self.n = self.hi - self.lo
        self.agents_per_step = A if mode == "strong" else A * world     # of the whole job
        self.fuse = max(1, min(args.fuse, args.ring))
        if stub:
            # launcher / collective plumbing rehearsal on the CPU: the "solve" is a copy, nothing is measured
            self.ring = [dict(u_ref=torch.zeros((self.n, 2)), u_out=torch.empty((self.n, 2)),
                              status=torch.zeros((self.n,), dtype=torch.int32)) for _ in range(args.ring)]
            self.flt = self.table = None
            return
        from sccav_cbf_amd.batched import BatchTable
        # every step reads a batch that is not touched again for RING - 1 steps: say so
        self.flt = _filter(args.lanes, streamed=not args.no_stream_hint)
        self.ring = make_ring(seed0, device, args.ring, agents=A, layout=args.layout, slots=args.slots,
                              lo=self.lo, hi=self.hi)
        self.table = BatchTable(self.ring) if self.n else None
        self.launches = 0
        # the launch of a whole walk over the ring, bound once (the host's share of a twenty-step region counts)
        self.full = self.flt.bind_batches(self.table) if (self.table is not None and self.fuse == len(self.ring)) else None

    def run(self, k):
        """Steps 0 ... k - 1 of a walk over the ring, `fuse` steps per launch."""
        ring = self.ring
        if self.stub:
            for i in range(k):
                b = ring[i % len(ring)]
                b["u_out"].copy_(b["u_ref"])
            return
        if self.table is None:     # a rank without agents (more ranks than tiles)
            return
        done, R, F = 0, len(ring), self.fuse
        while done < k:
            first = done % R
            n = min(F, k - done, R - first)
            if F == 1:
                b = ring[first]
                self.flt.solve(b["state"], b["u_ref"], b["obs"], b["obs_type"], u_out=b["u_out"], status=b["status"])
            else:
                if n == R and self.full is not None:
                    self.full()
                else:
                    self.flt.solve_batches(self.table, first=first, count=n)
            self.launches += 1
            done += n

    def launches_for(self, k):
        R, F = len(self.ring), self.fuse
        return (k // R) * -(-R // F) + -(-(k % R) // F)

    def launch_name(self):
        if self.stub:
            return "cpu stub"
        if self.fuse == 1:
            return "one sccbf_solve_f32 launch per step, eager"
        return ("one sccbf_solve_batches_f32 launch per %d steps: the steps are independent batches, each with its own "
                "arrays, handed to the library as %d batch records (the bits of one sccbf_solve_f32 launch per step)"
                % (self.fuse, self.fuse))


def timed_region(args, reg, sync, dist, stream):
    """Ramp, W warm-up steps, barrier, K timed steps, barrier.  Returns wall s, wall s up to the end of the closing
    barrier, device ms (HIP events on the launch stream), ramp steps, launches in the region."""
    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    ramp_steps = 0
    ev0 = ev1 = None
    if not reg.stub:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # An idle GPU needs some ten milliseconds of work to reach its clocks, more than a short warm-up gives it.
        # The untimed ramp runs what the timed region will run (events, launches, synchronize), several times: the
        # first pass of anything through the runtime and the interpreter costs tens of microseconds that a region
        # of one launch would otherwise carry.
        k_ramp = args.steps if args.steps <= 10 * len(reg.ring) else 10 * len(reg.ring)
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            ev0.record(stream)
            reg.run(k_ramp)
            ev1.record(stream)
            sync()
            ramp_steps += k_ramp
    reg.run(args.warmup)
    barrier()
    reg.launches = 0
    t0 = time.perf_counter()
    if ev0 is not None:
        ev0.record(stream)
    reg.run(args.steps)
    if ev1 is not None:
        ev1.record(stream)
    # every rank stops its own clock when its own K steps are done (the maximum over the ranks is the job's time);
    # the closing barrier follows, and the time up to its end is reported beside it
    sync()
    wall = time.perf_counter() - t0
    barrier()
    wall_barrier = time.perf_counter() - t0
    dev_ms = wall * 1e3 if (reg.stub or reg.table is None) else ev0.elapsed_time(ev1)
    return wall, wall_barrier, dev_ms, ramp_steps, reg.launches


def over_ranks(dist, device, world, wall, wall_barrier):
    """Maximum over the ranks of both clocks, every rank's own time, and the collective library's count of the ranks."""
    if dist is None:
        return wall, wall_barrier, [wall], 1
    tw = torch.tensor([wall, wall_barrier], dtype=torch.float64, device=device)
    allw = torch.empty((2 * world,), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(allw, tw)
    one = torch.ones((1,), dtype=torch.int32, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    allw = allw.view(world, 2).cpu()
    return float(allw[:, 0].max()), float(allw[:, 1].max()), [float(x) for x in allw[:, 0]], int(one.item())


def gather_cost(dist, device, world, reg, sync, reps=20):
    """Outside the timed region: what collecting every rank's answers of one step costs (the solve itself never
    communicates): all_gather of u_out (n, 2) f32 and status (n,) i32, blocks padded to the largest."""
    from sccav_cbf_amd.sharded import shard_bounds
    n = reg.n
    if reg.mode == "strong":
        n = max(b - a for a, b in (shard_bounds(reg.agents_per_step, world, r, 64) for r in range(world)))
    u = torch.zeros((n, 2), dtype=torch.float32, device=device)
    s = torch.zeros((n,), dtype=torch.int32, device=device)
    if reg.n:
        u[:reg.n], s[:reg.n] = reg.ring[0]["u_out"], reg.ring[0]["status"]
    gu = torch.empty((world * n, 2), dtype=torch.float32, device=device)
    gs = torch.empty((world * n,), dtype=torch.int32, device=device)
    for i in range(3 + reps):
        if i == 3:
            sync()
            tg = time.perf_counter()
        dist.all_gather_into_tensor(gu, u)
        dist.all_gather_into_tensor(gs, s)
    sync()
    tg = torch.tensor([(time.perf_counter() - tg) / reps * 1e3], dtype=torch.float64, device=device)
    dist.all_reduce(tg, op=dist.ReduceOp.MAX)
    return float(tg.item())


def worker(args):
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
        # the 2^20-agent measurement on its own (what tools/prof_kernel.sh --large-only runs under rocprofv3)
        out = large_batch(_filter(args.lanes, streamed=not args.no_stream_hint), device, args.layout, args.slots,
                          reps=args.steps if args.steps < 1000 else 200, graph=not args.no_graph)
        out["large_only"] = True
        print(json.dumps(out), flush=True)
        return 0

    sync = (lambda: None) if stub else torch.cuda.synchronize
    stream = None if stub else torch.cuda.Stream(device)
    ctx = torch.cuda.stream(stream) if stream is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        reg = Region(args, args.scaling, rank, world, device, stub)
        wall, wall_barrier, dev_ms, ramp_steps, launches = timed_region(args, reg, sync, dist, stream)
        wall, wall_barrier, rank_walls, coll_ranks = over_ranks(dist, device, world, wall, wall_barrier)
        gather_ms = gather_cost(dist, device, world, reg, sync) if dist is not None else None

        # the other scaling mode beside it (N > 1): the same region on the other partition of the work
        other = None
        if world > 1 and not args.no_other_scaling:
            omode = "strong" if args.scaling == "weak" else "weak"
            del reg.ring, reg.table
            oreg = Region(args, omode, rank, world, device, stub)
            ow, owb, odev, _, ol = timed_region(args, oreg, sync, dist, stream)
            ow, owb, owalls, _ = over_ranks(dist, device, world, ow, owb)
            og = gather_cost(dist, device, world, oreg, sync) if dist is not None else None
            other = {"scaling": omode, "value": oreg.agents_per_step * args.steps / ow, "unit": "solves/s",
                     "ms_per_step": ow / args.steps * 1e3, "agents_per_rank_and_step": oreg.n,
                     "agents_per_step_whole_job": oreg.agents_per_step,
                     "ms_per_step_by_rank": [x / args.steps * 1e3 for x in owalls],
                     "ms_per_step_with_all_gather": ow / args.steps * 1e3 + (og or 0.0),
                     "gather_ms_per_step": og, "launches": ol}
        rc = 0
        if rank == 0:
            out = headline(args, world, wall, wall_barrier, dev_ms, ramp_steps, launches, rank_walls, coll_ranks,
                           gather_ms, dist, stub, A, M,
                           reg_info=(reg.mode, reg.n, reg.agents_per_step, reg.fuse, reg.launch_name()))
            if other is not None:
                out["other_scaling"] = other
            if coll_ranks != world:
                print("bench.py: the collective saw %d ranks, expected %d" % (coll_ranks, world), file=sys.stderr)
                rc = 1
            if world == 1 and not stub and not args.headline_only:
                extras(args, out, reg, device, stream)
            if _saved_stdout is not None:
                sys.stdout.flush()
                os.dup2(_saved_stdout, 1)
            print(json.dumps(out), flush=True)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def headline(args, world, wall, wall_barrier, dev_ms, ramp_steps, launches, rank_walls, coll_ranks, gather_ms,
             dist, stub, A, M, reg_info):
    mode, n_rank, per_step, fuse, launch_name = reg_info
    value = per_step * args.steps / wall
    ms_step = wall / args.steps * 1e3
    ab_step = algorithmic_bytes(n_rank, M)                 # this rank's bytes per step
    kern_ms_step = dev_ms / args.steps
    launches = max(1, launches)
    layout_name = {"tiles": "tiles (A/64,M,2,64,4)", "planes": "planes (M,2,A,4)"}[args.layout]
    if mode == "strong":
        workload = ("C4 as stated: ONE batch of %d agents x 16 obstacle rows (8 ellipse + 8 collision cone) per step, "
                    "contiguous blocks of %d agents per GPU, DBM filter, alpha=1, R=diag(.5,.5), ring of %d distinct "
                    "batches" % (A, n_rank, args.ring))
    else:
        workload = ("C4: %d agents x 16 obstacle rows (8 ellipse + 8 collision cone) per GPU and step, DBM filter, alpha=1, "
                    "R=diag(.5,.5), ring of %d distinct batches" % (A, args.ring))
    out = {
        "metric": "CBF-QP solves/sec (2-input, 16-constraint)",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": mode, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "agents_per_gpu": n_rank, "agents_per_step_whole_job": per_step, "rows": M,
                   "lanes_per_qp": args.lanes, "obs_layout": layout_name,
                   "obs_slots": "prepared (kinds 4, 5: 1/a^2, 1/b^2, cos, sin and the cone's obstacle velocity are stored in "
                                "the slot, evaluated before the timed region)" if args.slots == "prepared"
                                else "as written (kinds 0, 1)",
                   "launch": launch_name, "batches_per_launch": fuse, "launches_in_timed_region": launches,
                   "clock_ramp_steps_before_warmup": ramp_steps},
    }
    if not stub:
        # achieved = algorithmic bytes of the timed region's launches / its device time (HIP events on the launch stream)
        achieved = ab_step / (kern_ms_step * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": _profile_json("traffic.json") if (fuse == RING and n_rank == AGENTS) else None,
            "kernel_ms": dev_ms / launches, "algorithmic_bytes_per_launch": ab_step * args.steps // launches,
            "batches_per_launch": args.steps / launches, "kernel_ms_per_step": kern_ms_step,
            "algorithmic_bytes_per_step": ab_step,
            "frac_by_wall_clock": frac_of_peak(ab_step, rank_walls[0] / args.steps * 1e3),
            "note": "achieved = algorithmic bytes of the timed region's launches / its device time by two HIP events on "
                    "the launch stream around the region; frac_by_wall_clock = the same bytes over this rank's host clock (ms_per_step at N = 1); "
                    "traffic = HBM bytes per launch by the FETCH_SIZE / WRITE_SIZE counters of a profiler run of this "
                    "command on these sources (profiles/), null when there is none"}
    else:
        out["data"] = "none (cpu stub: launcher rehearsal, nothing measured)"
        out["stub"] = True
        out["roofline"] = None
    if dist is not None:
        out["rccl_ranks" if not stub else "gloo_ranks"] = coll_ranks
        out["ms_per_step_by_rank"] = [x / args.steps * 1e3 for x in rank_walls]
        # the clock of ms_per_step stops when the rank's own steps are done; this one after the closing barrier
        out["ms_per_step_to_end_of_closing_barrier"] = wall_barrier / args.steps * 1e3
        out["gather_ms_per_step"] = gather_ms   # not part of value / ms_per_step
        out["ms_per_step_with_all_gather"] = ms_step + (gather_ms or 0.0)
    return out


def extras(args, out, reg, device, stream):
    """N = 1: what stands beside the headline.  Nothing here enters value / ms_per_step."""
    A, M = args.agents, ROWS
    flt, ring = reg.flt, reg.ring
    ab = algorithmic_bytes(A, M)
    status = torch.stack([b["status"] for b in ring]).cpu().numpy()
    counts = np.bincount(status.reshape(-1), minlength=3)
    out["status_counts"] = {"optimal": int(counts[0]), "infeasible": int(counts[1]), "nonfinite": int(counts[2])}
    # QPs whose rows admit no input: the filter reports status 1 and hands back the nominal input
    out["infeasible_fraction"] = float(counts[1]) / max(1, int(counts.sum()))
    r = out["roofline"]
    try:
        # the same steps one launch at a time (a serial hipGraph of sccbf_solve_f32 launches): what a caller whose
        # launches depend on each other gets; and the proof that the fused launch wrote the same answers
        got = [(b["u_out"].clone(), b["status"].clone()) for b in ring]
        steps = [_solve_fn(flt, b) for b in ring]
        serial_ms = _time_graph(steps, stream, 20)
        same = all(torch.equal(u, b["u_out"]) and torch.equal(st, b["status"]) for (u, st), b in zip(got, ring))
        r["one_at_a_time"] = {"kernel_ms": serial_ms, "achieved": ab / (serial_ms * 1e-3) / 1e9,
                              "frac": frac_of_peak(ab, serial_ms),
                              "note": "one sccbf_solve_f32 launch per step from a serial hipGraph: one kernel's own "
                                      "duration, what a chain of dependent launches gets"}
        r["fused_answers_equal_separate_launches"] = bool(same)
        # the same launches overlapping on three parallel branches of a hipGraph
        br_ms = _time_graph(steps * 5, stream, 20, branches=args.branches)
        r["branched_graph"] = {"ms_per_step": br_ms, "frac": frac_of_peak(ab, br_ms), "branches": args.branches,
                               "note": "one sccbf_solve_f32 launch per step, the launches of independent steps overlapping on "
                                       "parallel branches of one hipGraph"}
    except Exception as e:
        r["one_at_a_time"] = {"error": repr(e)}
    if args.steps < 5000:
        # a timed region of twenty steps is ONE launch between two synchronisations: the same measurement over
        # 4000 steps, outside the timed region, beside it
        try:
            n_st = 200
            ms = _events_ms(lambda: flt.solve_batches(reg.table), stream, n_st) / len(ring)
            ts = time.perf_counter()
            for _ in range(n_st):
                flt.solve_batches(reg.table)
            torch.cuda.synchronize()
            st_ms = (time.perf_counter() - ts) * 1e3 / (n_st * len(ring))
            out["steady_state"] = {"steps": n_st * len(ring), "ms_per_step": st_ms, "value": A / (st_ms * 1e-3),
                                   "ms_per_step_by_events": ms, "launch": reg.launch_name(),
                                   "roofline_frac": frac_of_peak(ab, st_ms),
                                   "note": "wall clock over 200 launches, outside the timed region"}
        except Exception as e:
            out["steady_state"] = {"error": repr(e)}
    if args.quick or A != AGENTS or args.layout != "tiles":
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.slots)
        return
    c4 = {"us_per_step": r["kernel_ms_per_step"] * 1e3, "solves_per_s": A / (r["kernel_ms_per_step"] * 1e-3),
          "frac": r["frac"], "dtype": "f32", "launch": "twenty batches per launch (the headline's timed region)"}
    if "one_at_a_time" in r and "frac" in r["one_at_a_time"]:
        c4["one_launch_per_step"] = {"us_per_step": r["one_at_a_time"]["kernel_ms"] * 1e3, "frac": r["one_at_a_time"]["frac"]}
    for key, fn in (
            # slots as the reference's objects hold them (kinds 0, 1): sin / cos / reciprocals per slot inside the launch
            ("slots_as_written", lambda: written_section(args, device, stream)),
            ("variants", lambda: variants_section(ring, device, stream)),
            ("c4_shard", lambda: shard_section(flt, ring, stream))):
        try:
            out[key] = fn()
        except Exception as e:  # extra information only
            out[key] = {"error": repr(e)}
    if isinstance(out.get("slots_as_written"), dict) and "one_launch_per_step" in out["slots_as_written"]:
        # what a closed loop gets: one dependent launch per tick, slots as written
        out["value_serial_written"] = out["slots_as_written"]["one_launch_per_step"]["solves_per_s"]
    del reg.ring, reg.table, ring
    torch.cuda.empty_cache()
    try:
        out["configs"] = configs_section(device, c4)
        cf = out["configs"]
        out["fp64"] = {"c4": cf["c4"]["f64"], "c5": cf["c5"]["f64"], "c2": cf["c2"]}
    except Exception as e:
        out["configs"] = {"error": repr(e)}
    try:
        torch.cuda.empty_cache()
        out["long_lists"] = long_lists_section(device)
    except Exception as e:
        out["long_lists"] = {"error": repr(e)}
    if not args.no_large:
        try:
            torch.cuda.empty_cache()
            out["roofline_large_batch"] = large_batch(flt, device, args.layout, args.slots)
        except Exception as e:
            out["roofline_large_batch"] = {"error": repr(e)}
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.slots)


def written_section(args, device, stream):
    flt = _filter(args.lanes, streamed=not args.no_stream_hint)
    ring = make_ring(1234 + 4, device, RING, slots="written")
    fused, single = fused_and_single(flt, ring, stream)
    return {"obs_slots": "as written (kinds 0, 1)", "twenty_batches_per_launch": _line(AGENTS, ROWS, fused),
            "one_launch_per_step": _line(AGENTS, ROWS, single)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=5000)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: a full 65536-agent batch per GPU and step; strong: one 65536-agent batch per step, "
                         "contiguous blocks of 65536 / N agents per GPU.  With N > 1 the line carries the other mode too "
                         "(other_scaling)")
    ap.add_argument("--no-other-scaling", action="store_true", help="N > 1: only the mode --scaling names")
    ap.add_argument("--fuse", type=int, default=RING,
                    help="steps per launch (sccbf_solve_batches_f32 over that many batch records); 1 = one sccbf_solve_f32 "
                         "launch per step")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per QP (0 = library default)")
    ap.add_argument("--agents", type=int, default=AGENTS)
    ap.add_argument("--no-graph", action="store_true", help="--large-only: launch eagerly instead of replaying a hipGraph")
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
                    help="parallel branches of the hipGraph in roofline.branched_graph")
    ap.add_argument("--ring", type=int, default=RING,
                    help="distinct batches the steps walk over (default 20: 718 MB, every step streams from HBM)")
    ap.add_argument("--quick", action="store_true",
                    help="N = 1: only the headline, its one-at-a-time / branched figures and the CPU baseline")
    ap.add_argument("--headline-only", action="store_true",
                    help="N = 1: nothing beside the headline (what the profiler scripts run: every kernel in the trace is "
                         "the timed region's)")
    ap.add_argument("--no-large", action="store_true", help="skip the extra 1M-agent measurement (N=1 only)")
    ap.add_argument("--large-only", action="store_true",
                    help="only the 2^20-agent measurement (with --steps < 1000: that many walks over its ring of 3)")
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
