"""Single-ego bridge from the object API to the C ABI (fp64, one agent per call).

The per-call cost is a few small host<->device copies; batches of agents should use
sccav_cbf_amd.BatchedCBF directly.  There is no CPU path: without a GPU these raise.
"""
import ctypes as C

import numpy as np

from .. import _lib


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _lib.SccbfError("the CBF filter needs an AMD GPU (HIP); there is no CPU fallback")
    return torch


def _pack(slots):
    """(M, 8) -> ABI layout (M, 2, 1, 4) for a single agent."""
    M = slots.shape[0]
    return np.ascontiguousarray(slots.reshape(M, 2, 1, 4))


_last_partials = (None, None)   # (key, result) of the latest partials() call


def partials(s, kinds, slots, zero_tol=1e-3):
    """h and partials of M obstacle slots for one ego state: (M, 6) = h, hx, hy, hth, hv, ht.
    The latest result is kept, keyed by the bits of every input: f(), dx(), dy(), dtheta(), dv(),
    dt() and gradient() on an unchanged obstacle list share one launch and one copy back."""
    global _last_partials
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    M = kinds.shape[0]
    if M == 0:
        return np.zeros((0, 6))
    s = np.ascontiguousarray(s, dtype=np.float64)
    slots = np.ascontiguousarray(slots, dtype=np.float64)
    key = (s.tobytes(), kinds.tobytes(), slots.tobytes(), float(zero_tol))
    if _last_partials[0] == key:
        return _last_partials[1].copy()
    torch = _torch()
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    st = torch.tensor(np.asarray(s, dtype=np.float64).reshape(1, 4), device=dev)
    ob = torch.tensor(_pack(np.asarray(slots, dtype=np.float64)), device=dev)
    ty = torch.tensor(kinds, device=dev)
    out = torch.empty((M, 2, 1, 4), dtype=torch.float64, device=dev)
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    p.zero_tol = zero_tol
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.sccbf_partials_f64(C.byref(p), 1, M, C.c_void_p(st.data_ptr()),
                                      C.c_void_p(ob.data_ptr()), C.c_void_p(ty.data_ptr()),
                                      C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
    o = out.cpu().numpy().reshape(M, 8)[:, :6].copy()
    _last_partials = (key, o)
    return o.copy()


def solve(params, s, u_ref, kinds, slots):
    """One QP for one ego: returns (u (2,), status, rows (M,4))."""
    torch = _torch()
    lib = _lib.load()
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    M = kinds.shape[0]
    dev = torch.device("cuda", torch.cuda.current_device())
    st = torch.tensor(np.asarray(s, dtype=np.float64).reshape(1, 4), device=dev)
    ur = torch.tensor(np.asarray(u_ref, dtype=np.float64).reshape(1, 2), device=dev)
    ob = torch.tensor(_pack(np.asarray(slots, dtype=np.float64)), device=dev)
    ty = torch.tensor(kinds, device=dev)
    u = torch.empty((1, 2), dtype=torch.float64, device=dev)
    status = torch.empty((1,), dtype=torch.int32, device=dev)
    rows = torch.empty((M, 1, 4), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.sccbf_solve_f64(C.byref(params), 1, M, C.c_void_p(st.data_ptr()),
                                   C.c_void_p(ur.data_ptr()), C.c_void_p(ob.data_ptr()),
                                   C.c_void_p(ty.data_ptr()), C.c_void_p(u.data_ptr()),
                                   C.c_void_p(status.data_ptr()), C.c_void_p(rows.data_ptr()),
                                   C.c_void_p(stream)))
    return u.cpu().numpy()[0], int(status.item()), rows.cpu().numpy()[:, 0, :]


def solve_sa(params, dt, s, slip, u_ref, kinds, slots):
    """One steer-rate QP for one ego (sccbf_solve_sa_f64): returns (u (2,), status, rows (M,4),
    slip (2,) = beta, beta_ref_last after the call)."""
    torch = _torch()
    lib = _lib.load()
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    M = kinds.shape[0]
    dev = torch.device("cuda", torch.cuda.current_device())
    st = torch.tensor(np.asarray(s, dtype=np.float64).reshape(1, 4), device=dev)
    sl = torch.tensor(np.asarray(slip, dtype=np.float64).reshape(1, 2), device=dev)
    ur = torch.tensor(np.asarray(u_ref, dtype=np.float64).reshape(1, 2), device=dev)
    ob = torch.tensor(_pack(np.asarray(slots, dtype=np.float64)), device=dev)
    ty = torch.tensor(kinds, device=dev)
    u = torch.empty((1, 2), dtype=torch.float64, device=dev)
    status = torch.empty((1,), dtype=torch.int32, device=dev)
    rows = torch.empty((M, 1, 4), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.sccbf_solve_sa_f64(C.byref(params), float(dt), 1, M, C.c_void_p(st.data_ptr()),
                                      C.c_void_p(sl.data_ptr()), C.c_void_p(ur.data_ptr()),
                                      C.c_void_p(ob.data_ptr()), C.c_void_p(ty.data_ptr()),
                                      C.c_void_p(u.data_ptr()), C.c_void_p(status.data_ptr()),
                                      C.c_void_p(rows.data_ptr()), C.c_void_p(stream)))
    return u.cpu().numpy()[0], int(status.item()), rows.cpu().numpy()[:, 0, :], sl.cpu().numpy()[0]
