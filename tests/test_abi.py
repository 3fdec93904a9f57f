"""The C-ABI library: loads, exports every declared symbol, validates arguments on the host
(no kernel is launched here, so this runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from sccav_cbf_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if _build.needs_build():
        _build.build_library()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sccbf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sccbf_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_symbols()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_version_and_strerror(lib):
    assert lib.sccbf_version() == _lib.ABI_VERSION
    assert lib.sccbf_strerror(0) == b"ok"
    for code in range(-7, 0):
        assert len(lib.sccbf_strerror(code)) > 3
    assert lib.sccbf_strerror(-99) == b"unknown error"


def test_params_struct_layout_and_defaults(lib):
    assert C.sizeof(_lib.Params) == 9 * 8 + 4 * 4
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    assert (p.alpha, p.lr, p.lf, p.L, p.zero_tol) == (1.0, 1.45, 1.45, 2.9, 1e-3)
    assert list(p.R) == [1.0, 0.0, 0.0, 1.0]
    assert (p.dynamics, p.lanes_per_qp, p.flags) == (0, 0, 0)


def test_host_side_argument_validation(lib):
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    nul = C.c_void_p(0)
    args = (nul,) * 8
    assert lib.sccbf_solve_f32(None, 1, 1, *args) == -1                 # no params
    assert lib.sccbf_solve_f32(C.byref(p), -1, 1, *args) == -1          # negative batch
    assert lib.sccbf_solve_f32(C.byref(p), 1, _lib.MAX_ROWS + 1, *args) == -2
    assert lib.sccbf_solve_f64(C.byref(p), 4, 2, *args) == -1           # null data pointers
    assert lib.sccbf_solve_f64(C.byref(p), 0, 2, *args) == 0            # empty batch: nothing to do
    assert lib.sccbf_qp_f32(C.byref(p), 0, 3, nul, nul, nul, nul, nul) == 0
    assert lib.sccbf_partials_f64(C.byref(p), 0, 3, nul, nul, nul, nul, nul) == 0
    p.R[0], p.R[3] = 1.0, -1.0
    assert lib.sccbf_solve_f32(C.byref(p), 1, 1, *args) == -3           # R not positive definite
    p.R[3] = 1.0
    p.R[1] = p.R[2] = 2.0
    assert lib.sccbf_solve_f32(C.byref(p), 1, 1, *args) == -3
    p.R[1] = p.R[2] = 0.0
    p.dynamics = 7
    assert lib.sccbf_solve_f32(C.byref(p), 1, 1, *args) == -1
    p.dynamics = 0
    p.lr = 0.0
    assert lib.sccbf_solve_f32(C.byref(p), 1, 1, *args) == -1


def test_nominal_and_vehicle_step_argument_validation(lib):
    assert C.sizeof(_lib.NominalParams) == 8 * 8 + 2 * 4
    q = _lib.NominalParams()
    lib.sccbf_default_nominal_params(C.byref(q))
    assert (q.axle, q.k, q.ks, q.kp, q.ki, q.kd, q.dt) == (2.0, 0.5, 0.01, 1.0, 0.0, 0.0, 0.1)
    assert abs(q.target_speed - 30.0 / 3.6) < 1e-15 and q.flags == 0
    nul = C.c_void_p(0)
    assert lib.sccbf_nominal_f32(None, 1, 1, nul, nul, nul, nul, nul, nul) == -1
    assert lib.sccbf_nominal_f32(C.byref(q), 1, 0, nul, nul, nul, nul, nul, nul) == -1     # empty course
    assert lib.sccbf_nominal_f64(C.byref(q), 0, 5, nul, nul, nul, nul, nul, nul) == 0      # empty batch
    assert lib.sccbf_nominal_f64(C.byref(q), 3, 5, nul, nul, nul, nul, nul, nul) == -1     # null data
    assert lib.sccbf_vehicle_step_f32(0, nul, nul, 1.45, 1.45, 0.5, 0.1, nul, nul) == 0
    assert lib.sccbf_vehicle_step_f32(2, nul, nul, 1.45, 1.45, 0.5, 0.1, nul, nul) == -1
    assert lib.sccbf_vehicle_step_f64(2, nul, nul, 0.0, 1.45, 0.5, 0.1, nul, nul) == -1    # lr must be > 0


def test_rollout_argument_validation(lib):
    assert C.sizeof(_lib.RolloutParams) == 2 * 8 + 2 * 4 + 2 * 8
    r = _lib.RolloutParams()
    lib.sccbf_default_rollout_params(C.byref(r))
    assert abs(r.max_steer - 0.5235987755982988) < 1e-15 and r.dt == 0.1 and (r.ticks, r.motion) == (0, 0)
    assert (r.seek_gain, r.seek_v_min) == (0.2, 3.0)      # radial_dynamic_obstacles.py:193
    p, q = _lib.Params(), _lib.NominalParams()
    lib.sccbf_default_params(C.byref(p))
    lib.sccbf_default_nominal_params(C.byref(q))
    nul = (C.c_void_p(0),) * 12
    r.ticks = 5
    call = lambda A, M, N: lib.sccbf_rollout_f32(C.byref(p), C.byref(q), C.byref(r), A, M, N, *nul)
    assert call(0, 4, 100) == 0                      # empty batch
    assert call(8, 4, 100) == -1                     # null data
    assert call(8, 4, 2049) == -1                    # a long course is fine (staged pass by pass): null data
    assert call(8, 65, 100) == -1 and call(8, 1025, 100) == -2   # a wavefront per agent beyond 64 slots: null data
    assert call(8, 0, 100) == -1 and call(8, 4, -1) == -1
    assert call(0, 4, 0) == 0                        # no course: the nominal input is zero
    r.motion = 9
    assert call(8, 4, 100) == -1
    r.motion, r.seek_gain = 2, -1.0
    assert call(8, 4, 100) == -1                     # seeker law needs gains >= 0
    r.motion, r.seek_gain = 0, 0.2
    for dyn in (1, 2, 3):
        p.dynamics = dyn
        assert call(0, 4, 100) == 0                  # every model has its loop
    p.dynamics, p.L = 1, 0.0
    assert call(0, 4, 100) == -1                     # the kinematic step needs the wheel base
    p.dynamics, p.L, p.flags = 0, 2.9, 4
    assert call(0, 4, 100) == -1                     # steering angles, not slip angles
    p.flags = 0
    assert lib.sccbf_rollout_f64(None, C.byref(q), C.byref(r), 8, 4, 100, *nul) == -1


def test_model_variants_argument_validation(lib):
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    nul = (C.c_void_p(0),) * 8
    p.dynamics = _lib.DYN_SADBM
    assert lib.sccbf_solve_f64(C.byref(p), 8, 4, *nul) == -1           # the steer-rate model needs its slip state
    p.dynamics = _lib.DYN_DUM
    assert lib.sccbf_solve_f64(C.byref(p), 0, 4, *nul) == 0
    assert lib.sccbf_solve_f64(C.byref(p), 8, 4, *nul) == -1           # null data
    p.dynamics = 7
    assert lib.sccbf_solve_f64(C.byref(p), 0, 4, *nul) == -1
    p.dynamics = 0                                                       # ignored by sccbf_solve_sa_*
    nul9 = (C.c_void_p(0),) * 9
    sa = lambda dt, A, M: lib.sccbf_solve_sa_f32(C.byref(p), dt, A, M, *nul9)
    assert sa(0.1, 0, 4) == 0
    assert sa(0.1, 8, 4) == -1                                           # null data
    assert sa(0.0, 0, 4) == -1 and sa(float("nan"), 0, 4) == -1 and sa(-1.0, 0, 4) == -1
    assert sa(0.1, 0, 0) == -1                                           # an empty list raises in the reference
    assert sa(0.1, 0, 1025) == -2
    p.flags = _lib.FLAG_SLIP_IO
    assert sa(0.1, 0, 4) == -1
    p.flags = 0
    assert lib.sccbf_solve_sa_f64(None, 0.1, 0, 4, *nul9) == -1


def test_solve_step_argument_validation(lib):
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    nul8 = (C.c_void_p(0),) * 8
    ss = lambda ms, dt, A, M: lib.sccbf_solve_step_f32(C.byref(p), ms, dt, A, M, *nul8)    # noqa: E731
    assert ss(0.5, 0.1, 0, 4) == 0
    assert ss(0.5, 0.1, 8, 4) == -1                                      # null data
    assert ss(-0.1, 0.1, 0, 4) == -1 and ss(float("nan"), 0.1, 0, 4) == -1 and ss(0.5, float("inf"), 0, 4) == -1
    assert ss(0.5, 0.1, 0, 1025) == -2
    for dyn in (_lib.DYN_KBM, _lib.DYN_DUM, _lib.DYN_SADBM):             # the step is the DBM drivers' update_com
        p.dynamics = dyn
        assert ss(0.5, 0.1, 0, 4) == -1
    p.dynamics = _lib.DYN_DBM
    p.flags = _lib.FLAG_SLIP_IO                                          # the step takes a steering angle
    assert ss(0.5, 0.1, 0, 4) == -1
    p.flags = 0
    assert lib.sccbf_solve_step_f64(None, 0.5, 0.1, 0, 4, *nul8) == -1


def test_ingest_and_lane_fit_argument_validation(lib):
    nul = C.c_void_p(0)
    ing = lambda A, M, buf=0.5, flags=0: lib.sccbf_ingest_bbox_f32(A, M, nul, nul, buf, flags, nul, nul)
    assert ing(0, 4) == 0 and ing(8, 0) == 0
    assert ing(8, 4) == -1                                   # null data
    assert ing(-1, 4) == -1 and ing(8, 1025) == -1 and ing(8, 4, float("nan")) == -1
    assert ing(0, 4, 0.5, _lib.FLAG_OBS_SHARED) == -1        # only the per-agent-kind and tiled flags apply
    fit = lambda A, K, deg, m, M, flags=0: lib.sccbf_lane_fit_f64(A, K, deg, nul, nul, nul, 1.5, m, M, flags, nul, nul)
    assert fit(0, 8, 3, 0, 2) == 0
    assert fit(8, 8, 3, 0, 2) == -1                          # null data
    assert fit(0, 0, 3, 0, 2) == -1 and fit(0, 8, 4, 0, 2) == -1 and fit(0, 8, -1, 0, 2) == -1
    assert fit(0, 8, 3, 2, 2) == -1 and fit(0, 8, 3, -1, 2) == -1 and fit(0, 8, 3, 0, 0) == -1
    assert fit(0, 8, 3, 0, 2, _lib.FLAG_TYPE_PER_AGENT) == -1


def test_no_cpu_fallback_in_front_end():
    """Without a GPU the batched front-end must refuse, not compute somewhere else."""
    import numpy as np
    import torch
    from sccav_cbf_amd.batched import BatchedCBF
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    f = BatchedCBF(R=np.eye(2))
    z = torch.zeros((1, 4))
    with pytest.raises(_lib.SccbfError):
        f.solve(z, torch.zeros((1, 2)), torch.zeros((1, 2, 1, 4)), torch.zeros((1,), dtype=torch.uint8))
    from sccav_cbf_amd.nominal import BatchedNominal, vehicle_step
    with pytest.raises(_lib.SccbfError):
        BatchedNominal().control(z, torch.zeros((3, 4)), torch.zeros((1,), dtype=torch.int32))
    with pytest.raises(_lib.SccbfError):
        vehicle_step(z, torch.zeros((1, 2)))


def test_batches_argument_validation(lib):
    assert C.sizeof(_lib.Batch) == 64
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    nul = C.c_void_p(0)
    for fn in (lib.sccbf_solve_batches_f32, lib.sccbf_solve_batches_f64):
        assert fn(None, 1, 16, 64, nul, None) == -1                      # no params
        assert fn(C.byref(p), -1, 16, 64, nul, None) == -1               # negative count
        assert fn(C.byref(p), _lib.MAX_BATCHES + 1, 16, 64, nul, None) == -1
        assert fn(C.byref(p), 4, 16, -1, nul, None) == -1
        assert fn(C.byref(p), 4, 16, 1 << 31, nul, None) == -1           # beyond the kernels' index range
        assert fn(C.byref(p), 4, _lib.MAX_ROWS + 1, 64, nul, None) == -2
        assert fn(C.byref(p), 0, 16, 64, nul, None) == 0                 # no batches: nothing to do
        assert fn(C.byref(p), 4, 16, 0, nul, None) == 0                  # no agents: nothing to do
        assert fn(C.byref(p), 4, 16, 64, nul, None) == -1                # no table
        for dyn in (_lib.DYN_DUM, _lib.DYN_SADBM):                       # the models that take more than these arrays
            p.dynamics = dyn
            assert fn(C.byref(p), 0, 16, 64, nul, None) == -1
        p.dynamics = 0
        p.flags = _lib.FLAG_CONE_BETA_STATE
        assert fn(C.byref(p), 0, 16, 64, nul, None) == -1
        p.flags = 0


def test_static_batches_argument_validation(lib):
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    nul = C.c_void_p(0)
    fn = lib.sccbf_solve_batches_static
    recs = (_lib.Batch * 3)()
    assert fn(C.byref(p), 2, 1, 16, recs, None) == -1                    # neither float nor double
    assert fn(None, 4, 1, 16, recs, None) == -1
    assert fn(C.byref(p), 4, -1, 16, recs, None) == -1
    assert fn(C.byref(p), 4, _lib.MAX_STATIC_BATCHES + 1, 16, recs, None) == _lib.ERR_BATCHES
    assert fn(C.byref(p), 8, 3, 16, nul, None) == -1                     # no records
    assert fn(C.byref(p), 8, 0, 16, nul, None) == 0                      # no batches: nothing to do
    assert fn(C.byref(p), 8, 3, 16, recs, None) == 0                     # three empty batches: nothing to do
    assert fn(C.byref(p), 8, 3, _lib.MAX_ROWS + 1, recs, None) == -2
    recs[1].A = 8                                                        # agents without arrays
    assert fn(C.byref(p), 4, 3, 16, recs, None) == -1
    recs[1].A = -1
    assert fn(C.byref(p), 4, 3, 16, recs, None) == -1
    recs[1].A = 1 << 31
    assert fn(C.byref(p), 4, 3, 16, recs, None) == -1
    recs[1].A = 0
    p.dynamics = _lib.DYN_DUM
    assert fn(C.byref(p), 4, 3, 16, recs, None) == -1


def test_min_violation_flag_where_it_is_refused(lib):
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    p.flags = _lib.FLAG_INFEASIBLE_MIN_VIOLATION
    nul = C.c_void_p(0)
    one = C.c_void_p(8)    # never dereferenced: every call below returns before a launch
    assert lib.sccbf_solve_f64(C.byref(p), 4, 2, one, one, one, one, one, one, nul, None) == -1      # needs rows_out
    assert lib.sccbf_solve_f64(C.byref(p), 0, 2, nul, nul, nul, nul, nul, nul, nul, None) == 0       # nothing to do
    assert lib.sccbf_solve_step_f32(C.byref(p), 0.5, 0.1, 0, 2, *([nul] * 8)) == -1
    assert lib.sccbf_solve_sa_f32(C.byref(p), 0.1, 0, 2, *([nul] * 10)) == -1
    assert lib.sccbf_solve_batches_f32(C.byref(p), 0, 2, 4, nul, None) == -1
    q, r = _lib.NominalParams(), _lib.RolloutParams()
    lib.sccbf_default_nominal_params(C.byref(q))
    lib.sccbf_default_rollout_params(C.byref(r))
    r.ticks = 1
    assert lib.sccbf_rollout_f32(C.byref(p), C.byref(q), C.byref(r), 0, 2, 0, *([nul] * 12)) == -1
