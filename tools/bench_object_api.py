#!/usr/bin/env python3
"""Latency of the reference-style object API on the GPU path: one ego, one call per tick, as the drivers use it
(stanley_controller_ellipse.py:733-747): DBM_CBF_2DS with 1, 4 and 16 obstacles, and the partials of a list.
    python tools/bench_object_api.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd.cbf.cbf import DBM_CBF_2DS  # noqa: E402
from sccav_cbf_amd.cbf.obstacles import CollisionCone2D  # noqa: E402


def main():
    if len(sys.argv) > 1:      # another bridge module to measure against (a file with partials / solve / solve_sa)
        import importlib.util
        from sccav_cbf_amd.cbf import _device
        spec = importlib.util.spec_from_file_location("other_bridge", sys.argv[1])
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = "sccav_cbf_amd.cbf"
        spec.loader.exec_module(mod)
        _device.partials, _device.solve, _device.solve_sa = mod.partials, mod.solve, mod.solve_sa
    rng = np.random.default_rng(0)
    for M in (1, 4, 16):
        cbf = DBM_CBF_2DS(alpha=1)
        cbf.set_model_params(lr=1.45, lf=1.45)
        cbf.set_qp_cost_weight(np.diag([0.5, 0.5]))
        ts, tp = [], []
        for it in range(400):
            s = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-0.3, 0.3), rng.uniform(5, 10)])
            obst = {i: CollisionCone2D(3.0, s, np.array([30.0 + 5 * i, rng.uniform(-10, 10), 0.0, 2.0])) for i in range(M)}
            t0 = time.perf_counter()
            cbf.obstacle_list2d.update(obst)
            cbf.update_state(s=s)
            u = cbf.solve_cbf(np.array([0.5, 0.02]))
            t1 = time.perf_counter()
            h = cbf.obstacle_list2d.f()
            g = cbf.obstacle_list2d.gradient()
            t2 = time.perf_counter()
            ts.append(t1 - t0)
            tp.append(t2 - t1)
        ts, tp = np.array(ts[50:]) * 1e6, np.array(tp[50:]) * 1e6
        print("M = %2d: update + solve_cbf %.0f us per tick (median; min %.0f); f() + gradient() of the list %.0f us" %
              (M, np.median(ts), ts.min(), np.median(tp)), flush=True)
    assert u is not None and h is not None and g is not None


if __name__ == "__main__":
    main()
