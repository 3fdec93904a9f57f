"""MI355X-native batched CBF-QP safety filter (hot path of sccav_cbf) behind a C ABI."""
from ._lib import (DYN_DBM, DYN_KBM, FLAG_OBS_SHARED, FLAG_TYPE_PER_AGENT, MAX_ROWS, OBS_CONE,
                   OBS_ELLIPSE, OBS_EMPTY, OBS_LANE, OBS_RADIAL, STATUS_INFEASIBLE,
                   STATUS_NONFINITE, STATUS_OPTIMAL, SccbfError)

__all__ = ["BatchedCBF", "BatchedNominal", "ClosedLoop", "vehicle_step", "SccbfError", "OBS_ELLIPSE", "OBS_CONE", "OBS_LANE", "OBS_RADIAL",
           "OBS_EMPTY", "DYN_DBM", "DYN_KBM", "STATUS_OPTIMAL", "STATUS_INFEASIBLE",
           "STATUS_NONFINITE", "FLAG_OBS_SHARED", "FLAG_TYPE_PER_AGENT", "MAX_ROWS"]


def __getattr__(name):
    # torch is imported only when the batched front-end is actually used
    if name == "BatchedCBF":
        from .batched import BatchedCBF
        return BatchedCBF
    if name in ("BatchedNominal", "ClosedLoop", "vehicle_step"):
        from . import nominal
        return getattr(nominal, name)
    raise AttributeError(name)
