"""ctypes binding of libsccbf.so (the C ABI declared in include/sccbf.h).

There is no CPU fallback: if the HIP library is missing, load() raises.
"""
import ctypes as C
import os

from . import _build
from ._build import LIB

ABI_VERSION = 300   # SCCBF_VERSION of include/sccbf.h this binding was written against

OBS_ELLIPSE, OBS_CONE, OBS_LANE, OBS_RADIAL, OBS_EMPTY = 0, 1, 2, 3, 255
OBS_ELLIPSE_PREP, OBS_CONE_PREP = 4, 5   # the same obstacles, per-obstacle constants already evaluated
OBS_ELLIPSE_BOX, OBS_CONE_BOX = 6, 7     # the slot is the obstacle's bounding box + buffer, the filter makes the obstacle
OBS_LANE6 = 8                            # a lane polynomial of degree up to 6: c0 ... c6, buffer
DYN_DBM, DYN_KBM, DYN_DUM, DYN_SADBM = 0, 1, 2, 3
STATUS_OPTIMAL, STATUS_INFEASIBLE, STATUS_NONFINITE = 0, 1, 2
FLAG_OBS_SHARED, FLAG_TYPE_PER_AGENT, FLAG_SLIP_IO, FLAG_OBS_TILED = 1, 2, 4, 8
FLAG_OBS_AGENT_MAJOR = 128
FLAG_CONE_BETA_STATE = 16
FLAG_OBS_STREAMED = 32
FLAG_INFEASIBLE_MIN_VIOLATION = 64
MAX_ROWS = 1024
ERR_ARG, ERR_ROWS, ERR_WEIGHT, ERR_LAUNCH, ERR_VARIANT, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6
ERR_BATCHES = -7

EXPORTS = ("sccbf_version", "sccbf_strerror", "sccbf_default_params", "sccbf_solve_f32",
           "sccbf_solve_f64", "sccbf_qp_f32", "sccbf_qp_f64", "sccbf_partials_f32",
           "sccbf_partials_f64", "sccbf_default_nominal_params", "sccbf_nominal_f32",
           "sccbf_nominal_f64", "sccbf_vehicle_step_f32", "sccbf_vehicle_step_f64",
           "sccbf_default_rollout_params", "sccbf_rollout_f32", "sccbf_rollout_f64",
           "sccbf_solve_sa_f32", "sccbf_solve_sa_f64", "sccbf_ingest_bbox_f32", "sccbf_ingest_bbox_f64",
           "sccbf_lane_fit_f32", "sccbf_lane_fit_f64", "sccbf_solve_step_f32", "sccbf_solve_step_f64",
           "sccbf_prepare_obs_f32", "sccbf_prepare_obs_f64", "sccbf_vehicle_step_kin_f32",
           "sccbf_vehicle_step_kin_f64", "sccbf_solve_batches_f32", "sccbf_solve_batches_f64",
           "sccbf_solve_batches_static", "sccbf_prewarm")
MOTION_STATIC, MOTION_CONSTANT_VELOCITY, MOTION_SEEKER = 0, 1, 2
STEP_KIN_ACC, STEP_KIN_VEL, STEP_UNICYCLE = 0, 1, 2
NOMINAL_WAYPOINT_SPEED = 1


class Params(C.Structure):
    """struct sccbf_params of include/sccbf.h."""
    _fields_ = [("alpha", C.c_double), ("lr", C.c_double), ("lf", C.c_double), ("L", C.c_double),
                ("R", C.c_double * 4), ("zero_tol", C.c_double), ("dynamics", C.c_int32),
                ("lanes_per_qp", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_int32)]


class NominalParams(C.Structure):
    """struct sccbf_nominal_params of include/sccbf.h."""
    _fields_ = [("axle", C.c_double), ("k", C.c_double), ("ks", C.c_double), ("kp", C.c_double),
                ("ki", C.c_double), ("kd", C.c_double), ("dt", C.c_double), ("target_speed", C.c_double),
                ("flags", C.c_uint32), ("reserved", C.c_int32)]


class RolloutParams(C.Structure):
    """struct sccbf_rollout_params of include/sccbf.h."""
    _fields_ = [("max_steer", C.c_double), ("dt", C.c_double), ("ticks", C.c_int32), ("motion", C.c_int32),
                ("seek_gain", C.c_double), ("seek_v_min", C.c_double)]


class Batch(C.Structure):
    """struct sccbf_batch of include/sccbf.h: one batch of sccbf_solve_batches_* (64 bytes)."""
    _fields_ = [("A", C.c_int64), ("state", C.c_void_p), ("u_ref", C.c_void_p), ("obs", C.c_void_p),
                ("obs_type", C.c_void_p), ("u_out", C.c_void_p), ("status", C.c_void_p), ("rows_out", C.c_void_p)]


MAX_BATCHES = 65535
MAX_STATIC_BATCHES = 1024


class SccbfError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process, and it has to be the one the tensors come from: PyTorch ships its own copy, so
    # torch is imported before the library is mapped.  Mapped first, libsccbf.so would pull in the system's
    # runtime, torch would then bring its own, and the library's kernels would fail to launch on torch's streams.
    try:
        import torch  # noqa: F401
    except ImportError:   # a caller without PyTorch (ctypes on its own hipMalloc memory): the system's runtime
        pass
    alt = os.environ.get("SCCBF_LIBRARY")   # another build of the same sources (measurements of build options)
    if alt:
        return _bind(C.CDLL(alt))
    if not os.path.exists(LIB):
        raise SccbfError(
            "libsccbf.so not found at %s: build it with `python -m sccav_cbf_amd._build` "
            "(hipcc, gfx950). There is no CPU fallback." % LIB)
    if _build.needs_build():
        # a library older than its sources may no longer match the argument lists bound below
        if os.environ.get("SCCBF_NO_REBUILD"):
            raise SccbfError("libsccbf.so is older than its sources (csrc/, include/sccbf.h): rebuild with "
                             "`python -m sccav_cbf_amd._build`")
        _build.build_library()
    return _bind(C.CDLL(LIB))


def _bind(L):
    global _lib
    vp = C.c_void_p
    L.sccbf_version.restype = C.c_int
    if L.sccbf_version() != ABI_VERSION:
        raise SccbfError("libsccbf.so reports ABI version %d, this binding expects %d: rebuild with "
                         "`python -m sccav_cbf_amd._build`" % (L.sccbf_version(), ABI_VERSION))
    L.sccbf_strerror.restype = C.c_char_p
    L.sccbf_strerror.argtypes = [C.c_int]
    L.sccbf_default_params.restype = None
    L.sccbf_default_params.argtypes = [C.POINTER(Params)]
    for name in ("sccbf_solve_f32", "sccbf_solve_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(Params), C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    for name in ("sccbf_solve_sa_f32", "sccbf_solve_sa_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(Params), C.c_double, C.c_int64, C.c_int32] + [vp] * 9
    for name in ("sccbf_ingest_bbox_f32", "sccbf_ingest_bbox_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.c_int64, C.c_int32, vp, vp, C.c_double, C.c_uint32, vp, vp]
    for name in ("sccbf_prepare_obs_f32", "sccbf_prepare_obs_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.c_int64, C.c_int32, vp, vp, C.c_uint32, vp]
    for name in ("sccbf_lane_fit_f32", "sccbf_lane_fit_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, C.c_double, C.c_int32, C.c_int32, C.c_uint32,
                      vp, vp]
    for name in ("sccbf_qp_f32", "sccbf_qp_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(Params), C.c_int64, C.c_int32, vp, vp, vp, vp, vp]
    for name in ("sccbf_partials_f32", "sccbf_partials_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(Params), C.c_int64, C.c_int32, vp, vp, vp, vp, vp]
    L.sccbf_default_nominal_params.restype = None
    L.sccbf_default_nominal_params.argtypes = [C.POINTER(NominalParams)]
    for name in ("sccbf_nominal_f32", "sccbf_nominal_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(NominalParams), C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp]
    for name in ("sccbf_vehicle_step_f32", "sccbf_vehicle_step_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.c_int64, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp]
    for name in ("sccbf_solve_step_f32", "sccbf_solve_step_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(Params), C.c_double, C.c_double, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.sccbf_default_rollout_params.restype = None
    L.sccbf_default_rollout_params.argtypes = [C.POINTER(RolloutParams)]
    for name in ("sccbf_rollout_f32", "sccbf_rollout_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(Params), C.POINTER(NominalParams), C.POINTER(RolloutParams), C.c_int64,
                      C.c_int32, C.c_int32] + [vp] * 12
    for name in ("sccbf_vehicle_step_kin_f32", "sccbf_vehicle_step_kin_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.c_int64, vp, vp, C.c_int32, C.c_double, C.c_double, C.c_double, vp]
    for name in ("sccbf_solve_batches_f32", "sccbf_solve_batches_f64",
           "sccbf_solve_batches_static", "sccbf_prewarm"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(Params), C.c_int32, C.c_int32, C.c_int64, vp, vp]
    L.sccbf_prewarm.restype = C.c_int
    L.sccbf_prewarm.argtypes = []
    L.sccbf_solve_batches_static.restype = C.c_int
    L.sccbf_solve_batches_static.argtypes = [C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise SccbfError("sccbf: %s (code %d)" % (load().sccbf_strerror(rc).decode(), rc))
