"""Single-ego bridge from the object API to the C ABI (fp64, one agent per call).

One call = one copy in, one launch, one copy out: inputs and outputs of a list of M obstacles live in one
device buffer that is kept per (device, M).  Batches of agents should use sccav_cbf_amd.BatchedCBF
directly.  There is no CPU path: without a GPU these raise.  Like the reference's objects, not re-entrant.
"""
import ctypes as C

import numpy as np

from .. import _lib


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _lib.SccbfError("the CBF filter needs an AMD GPU (HIP); there is no CPU fallback")
    return torch


def _r4(n):
    return (n + 3) // 4 * 4


class _Bridge:
    """Device buffer of one ego with M obstacle slots, in doubles, every part on a 32-byte boundary:
       state 4 | u_ref 2 | obs 8 M (the ABI's (M, 2, 1, 4)) | kinds M bytes | slip 2 | u 2 | status (int32) | out 8 M
    Everything up to and including slip is uploaded in one copy, everything from slip on comes back in one."""

    def __init__(self, torch, dev, M):
        self.torch, self.M = torch, M
        self.o_state, self.o_uref, self.o_obs = 0, 4, 8
        self.o_kind = 8 + 8 * M
        self.o_slip = self.o_kind + _r4((M + 7) // 8)
        self.o_u = self.o_slip + 4
        self.o_status = self.o_u + 4
        self.o_out = self.o_status + 4
        self.buf = torch.zeros(self.o_out + 8 * M, dtype=torch.float64, device=dev)
        self.host = np.zeros(self.o_u)
        self.kinds = self.host[self.o_kind:self.o_slip].view(np.uint8)

    def ptr(self, off):
        return C.c_void_p(self.buf.data_ptr() + 8 * off)

    def upload(self, s, kinds, slots, u_ref=None, slip=None):
        h, M = self.host, self.M
        h[0:4] = np.asarray(s, dtype=np.float64).reshape(4)
        h[4:6] = 0.0 if u_ref is None else np.asarray(u_ref, dtype=np.float64).reshape(2)
        h[8:8 + 8 * M] = np.asarray(slots, dtype=np.float64).reshape(-1)
        self.kinds[:M] = kinds
        h[self.o_slip:self.o_slip + 2] = 0.0 if slip is None else np.asarray(slip, dtype=np.float64).reshape(2)
        self.buf[:self.o_u].copy_(self.torch.from_numpy(h))

    def download(self):
        """slip (2), u (2), status, out (8 M doubles: rows (M, 4) first, or partials (M, 2, 1, 4))."""
        d = self.buf[self.o_slip:].cpu().numpy()
        return d[0:2], d[4:6], int(d[8:9].view(np.int32)[0]), d[12:]


_bridges = {}


def _bridge(M):
    torch = _torch()
    key = (torch.cuda.current_device(), M)
    b = _bridges.get(key)
    if b is None:
        if len(_bridges) > 64:      # lists of many different lengths: keep the table small
            _bridges.clear()
        b = _bridges[key] = _Bridge(torch, torch.device("cuda", key[0]), M)
    return b


def _stream(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


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
    lib = _lib.load()
    b = _bridge(M)
    b.upload(s, kinds, slots)
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    p.zero_tol = zero_tol
    _lib.check(lib.sccbf_partials_f64(C.byref(p), 1, M, b.ptr(b.o_state), b.ptr(b.o_obs), b.ptr(b.o_kind),
                                      b.ptr(b.o_out), _stream(b.torch)))
    o = b.download()[3].reshape(M, 8)[:, :6].copy()
    _last_partials = (key, o)
    return o.copy()


def solve(params, s, u_ref, kinds, slots):
    """One QP for one ego: returns (u (2,), status, rows (M,4))."""
    lib = _lib.load()
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    M = kinds.shape[0]
    b = _bridge(M)
    b.upload(s, kinds, slots, u_ref=u_ref)
    _lib.check(lib.sccbf_solve_f64(C.byref(params), 1, M, b.ptr(b.o_state), b.ptr(b.o_uref), b.ptr(b.o_obs),
                                   b.ptr(b.o_kind), b.ptr(b.o_u), b.ptr(b.o_status), b.ptr(b.o_out),
                                   _stream(b.torch)))
    _, u, status, out = b.download()
    return u.copy(), status, out[:4 * M].reshape(M, 4).copy()


def solve_sa(params, dt, s, slip, u_ref, kinds, slots):
    """One steer-rate QP for one ego (sccbf_solve_sa_f64): returns (u (2,), status, rows (M,4),
    slip (2,) = beta, beta_ref_last after the call)."""
    lib = _lib.load()
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    M = kinds.shape[0]
    b = _bridge(M)
    b.upload(s, kinds, slots, u_ref=u_ref, slip=slip)
    _lib.check(lib.sccbf_solve_sa_f64(C.byref(params), float(dt), 1, M, b.ptr(b.o_state), b.ptr(b.o_slip),
                                      b.ptr(b.o_uref), b.ptr(b.o_obs), b.ptr(b.o_kind), b.ptr(b.o_u),
                                      b.ptr(b.o_status), b.ptr(b.o_out), _stream(b.torch)))
    sl, u, status, out = b.download()
    return u.copy(), status, out[:4 * M].reshape(M, 4).copy(), sl.copy()
