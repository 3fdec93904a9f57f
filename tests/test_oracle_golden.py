"""The oracle against the numbers the reference tree itself holds (CPU only)."""
import os

import numpy as np

from oracle import replay

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_course_fixture_matches_reference_run():
    """sce_course.npz was produced by the reference's own cubic_spline_planner
    (tests/golden/make_reference_fixtures.py); these values were read off that run."""
    cx, cy, cyaw = replay.load_course()
    assert len(cx) == 2034
    k = replay.scenario_constants(cx, cy)
    assert k["last_idx"] == 2033
    assert abs(k["o_cx"] - 75.63083239590846) < 1e-12
    assert abs(k["o_cy"] + 35.149858832676415) < 1e-12
    assert abs(cyaw[0] - 0.3551942145875729) < 1e-12
    assert abs(k["a_cone"] - 11.180339887498949) < 1e-12


def test_replay_reproduces_beta_vs_time_mat():
    """C1: the as-shipped closed loop (Stanley + P, one collision-cone row, alpha 1,
    R = diag(.5,.5)) must reproduce the 277 slip-angle samples that the reference saved
    with its cvxopt solver in test_scripts/beta_vs_time.mat.  Bar: 1e-3 degree."""
    t, beta_deg, log = replay.replay_stanley_cone()
    g = np.load(os.path.join(GOLDEN, "beta_vs_time.npz"))
    assert len(t) == len(g["t_arr"]) == 277
    assert np.allclose(t, g["t_arr"], atol=1e-9)
    diff = np.abs(beta_deg - g["beta_deg"])
    assert diff.max() <= 1e-3, diff.max()
    # the filter actually intervenes on this run (59 of 276 ticks)
    active = sum(1 for (_, a_ref, d_ref, a, d) in log if abs(a - a_ref) > 1e-9 or abs(d - d_ref) > 1e-9)
    assert active == 59
    # spot values recorded in the .mat
    assert np.allclose(g["beta_deg"][1:5], [-3.34196567, -3.28715643, -3.22429197, -2.46970969], atol=1e-7)


def test_replay_with_true_ellipse_row_stays_outside():
    """C1b: same loop with an Ellipse2D row (a=20, b=10); oracle-checked only: the ego must
    never enter the ellipse once the filter is active."""
    t, beta_deg, log = replay.replay_stanley_cone(row_kind="ellipse")
    cx, cy, _ = replay.load_course()
    k = replay.scenario_constants(cx, cy)
    h = [((s[0] - k["o_cx"]) / 20.0) ** 2 + ((s[1] - k["o_cy"]) / 10.0) ** 2 - 1 for (s, *_r) in log]
    assert min(h) > -1e-6
    assert len(t) > 100


def test_oracle_reproduces_its_frozen_vectors():
    """tests/golden/oracle_cases.npz (written by make_oracle_fixtures.py from this oracle): barrier values and
    partials of every obstacle kind with their degenerate cases, the filter of every vehicle model on two batches,
    hand-made and random QPs.  The oracle of today must give the same numbers: the GPU parity tests are measured
    against it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_oracle_fixtures", os.path.join(GOLDEN, "make_oracle_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    now = mod.build()
    g = np.load(os.path.join(GOLDEN, "oracle_cases.npz"))
    assert sorted(now) == sorted(g.files)
    for name in g.files:
        a, b = np.asarray(now[name]), g[name]
        assert a.shape == b.shape and a.dtype == b.dtype, name
        if a.dtype.kind == "f":
            assert np.allclose(a, b, rtol=1e-11, atol=1e-12, equal_nan=True), name
        else:
            assert np.array_equal(a, b), name
    # the degenerate cases are in the file and finite
    assert np.isfinite(g["bar_out"]).all() and len(g["bar_kind"]) >= 200
    assert (g["qp_hand_status"] == 1).sum() == 3 and (g["qp_rand_status"] == 1).sum() > 60
