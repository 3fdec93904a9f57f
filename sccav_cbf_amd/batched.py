"""Batched front-end of the CBF-QP safety filter: torch tensors in HBM -> C ABI -> HIP kernels.

One call = one control tick for every agent of the batch (the work that the
reference does per ego in DBM_CBF_2DS.solve_cbf, cbf/cbf.py:166-220, with the
obstacle partials of cbf/obstacles.py).  Tensors hold device memory and the
current stream; all arithmetic is in libsccbf.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import (DYN_DBM, DYN_KBM, FLAG_OBS_SHARED, FLAG_OBS_TILED, FLAG_SLIP_IO, FLAG_TYPE_PER_AGENT,
                   OBS_CONE, OBS_ELLIPSE,
                   OBS_EMPTY, OBS_LANE, OBS_RADIAL)

_DYN = {"dbm": DYN_DBM, "kbm": DYN_KBM, "dum": _lib.DYN_DUM, "sadbm": _lib.DYN_SADBM,
        DYN_DBM: DYN_DBM, DYN_KBM: DYN_KBM, _lib.DYN_DUM: _lib.DYN_DUM, _lib.DYN_SADBM: _lib.DYN_SADBM}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class BatchedCBF:
    """CBF-QP filter over a batch of A agents with M obstacle slots each.

    Parameters mirror the reference filter objects: alpha (set_alpha), lr/lf or L
    (set_model_params), R (set_qp_cost_weight); dynamics "dbm" = DBM_CBF_2DS
    (u = [a, delta]), "kbm" = KBM_VC_CBF2D (u = [v, delta]), "dum" = DUM_CBF_2DS
    (u = [a, omega], no steering transform) or "sadbm" = SADBM_CBF_2DS (u = [a, delta] in and
    out, the QP works on [a, d(beta)/dt]; solve() then needs `slip` and `dt`).
    lanes_per_qp: 0 lets the library choose; 1/2/4/8/16 cooperating lanes per QP
    with rows in registers; 16 beyond 128 rows, 32, 64 = rows in LDS only (64 = one wavefront per QP).
    """

    def __init__(self, alpha=1.0, lr=1.45, lf=1.45, L=2.9, R=None, dynamics="dbm",
                 zero_tol=1e-3, lanes_per_qp=0, slip_io=False, streamed_obs=False, min_violation=False):
        self._lib = _lib.load()
        self.params = _lib.Params()
        self._lib.sccbf_default_params(C.byref(self.params))
        self.params.alpha = float(alpha)
        self.params.lr, self.params.lf, self.params.L = float(lr), float(lf), float(L)
        self.params.zero_tol = float(zero_tol)
        self.params.dynamics = _DYN[dynamics]
        self.params.lanes_per_qp = int(lanes_per_qp)
        # slip_io: u_ref[:,1] / u_out[:,1] are slip angles (DBM) or yaw rates (KBM) already
        self.slip_io = bool(slip_io)
        # streamed_obs: every call brings obstacle slots that are not read again soon (a performance hint)
        self.streamed_obs = bool(streamed_obs)
        # min_violation: an agent whose rows admit no input (status 1) gets, instead of its nominal input, the input
        # whose largest row violation is least (SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION): solve() of the bicycle and unicycle
        # models and solve_qp(); solve() then always returns the rows as well (the second pass reads them)
        self.min_violation = bool(min_violation)
        self.set_qp_cost_weight(np.eye(2) if R is None else R)

    # -- setters with the reference's names (cbf/cbf.py:147-157)
    def set_alpha(self, alpha=1.0):
        self.params.alpha = float(alpha)

    def set_model_params(self, lr=None, lf=None, L=None):
        if lr is not None:
            self.params.lr = float(lr)
        if lf is not None:
            self.params.lf = float(lf)
        if L is not None:
            self.params.L = float(L)

    def set_qp_cost_weight(self, R):
        R = np.asarray(R, dtype=np.float64)
        if R.shape != (2, 2):
            raise ValueError("Expected a symmetric matrix of size 2 as input.")
        for i in range(4):
            self.params.R[i] = float(R.reshape(-1)[i])

    # -- the hot path
    def solve(self, state, u_ref, obs, obs_type, u_out=None, status=None, rows_out=None,
              want_rows=False, slip=None, dt=None, cone_beta_state=False, step_dt=None,
              max_steer=np.radians(30.0), beta_out=None):
        """state (A,4), u_ref (A,2), obs (M,2,A,4) [or (ceil(A/64),M,2,64,4): tiles of 64 agents,
        see pack_obs_tiled; or (A,M,2,4): per-agent lists, the layout for more than 128 slots, see
        planes_to_agents; or (M,2,4): one table shared by all agents], obs_type uint8 (M,)
        [or (M,A)].  Returns (u_out (A,2), status (A,) int32)
        and, with want_rows/rows_out, rows (M,A,4) = Lg_h0, Lg_h1, b, h.
        dynamics "sadbm": slip (A,2) = beta, beta_ref_last is read and updated in place, dt is the
        time step of this call; cone_beta_state makes every cone take the agent's beta
        (reference cbf/cbf.py:424-427).  step_dt (dynamics "dbm", not slip_io, no rows): the same
        launch then moves every agent by one vehicle step of step_dt with its filtered input, steering
        clipped to max_steer: state is overwritten, beta_out (A,) optional (nominal.vehicle_step).
        Asynchronous on the current stream."""
        if not state.is_cuda:
            raise _lib.SccbfError("BatchedCBF.solve needs tensors on the GPU (no CPU fallback)")
        sadbm = self.params.dynamics == _lib.DYN_SADBM
        if sadbm and (slip is None or dt is None):
            raise ValueError("the steer-rate model needs slip (A,2) and dt")
        if not sadbm and (slip is not None or cone_beta_state):
            raise ValueError("slip / cone_beta_state belong to the steer-rate model (dynamics='sadbm')")
        if step_dt is not None and (sadbm or want_rows or rows_out is not None):
            raise ValueError("step_dt goes with the plain solve of the DBM model")
        step = dt
        dt = state.dtype
        if dt not in (torch.float32, torch.float64):
            raise TypeError("state must be float32 or float64")
        A = state.shape[0]
        dev = state.device
        lflag, M, exp_obs = obs_layout(obs, A)
        flags = lflag | (FLAG_SLIP_IO if self.slip_io else 0) | (_lib.FLAG_OBS_STREAMED if self.streamed_obs else 0)
        if obs_type.dim() == 2:
            flags |= FLAG_TYPE_PER_AGENT
            exp_ty = (M, A)
        else:
            exp_ty = (M,)
        for name, t, shape, dtype in (("state", state, (A, 4), dt), ("u_ref", u_ref, (A, 2), dt),
                                      ("obs", obs, exp_obs, dt),
                                      ("obs_type", obs_type, exp_ty, torch.uint8)):
            if tuple(t.shape) != shape or t.dtype != dtype or t.device != dev or not t.is_contiguous():
                raise ValueError("%s: expected contiguous %s %s on %s, got %s %s on %s" % (
                    name, shape, dtype, dev, tuple(t.shape), t.dtype, t.device))
        if u_out is None:
            u_out = torch.empty((A, 2), dtype=dt, device=dev)
        if status is None:
            status = torch.empty((A,), dtype=torch.int32, device=dev)
        if self.min_violation:
            if sadbm or step_dt is not None:
                raise ValueError("min_violation goes with the plain solve of the bicycle and unicycle models")
            flags |= _lib.FLAG_INFEASIBLE_MIN_VIOLATION
            want_rows = True
        if want_rows and rows_out is None:
            rows_out = torch.empty((M, A, 4), dtype=dt, device=dev)
        for name, t, shape, dtype in (("u_out", u_out, (A, 2), dt), ("status", status, (A,), torch.int32),
                                      ("rows_out", rows_out, (M, A, 4), dt), ("slip", slip, (A, 2), dt),
                                      ("beta_out", beta_out, (A,), dt)):
            if t is not None and (tuple(t.shape) != shape or t.dtype != dtype or t.device != dev
                                  or not t.is_contiguous()):
                raise ValueError("%s: expected contiguous %s %s on %s" % (name, shape, dtype, dev))
        if cone_beta_state:
            flags |= _lib.FLAG_CONE_BETA_STATE
        self.params.flags = flags
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if sadbm:
                fn = self._lib.sccbf_solve_sa_f32 if dt == torch.float32 else self._lib.sccbf_solve_sa_f64
                rc = fn(C.byref(self.params), float(step), A, M, _ptr(state), _ptr(slip), _ptr(u_ref), _ptr(obs),
                        _ptr(obs_type), _ptr(u_out), _ptr(status), _ptr(rows_out), C.c_void_p(stream))
            elif step_dt is not None:
                fn = self._lib.sccbf_solve_step_f32 if dt == torch.float32 else self._lib.sccbf_solve_step_f64
                rc = fn(C.byref(self.params), float(max_steer), float(step_dt), A, M, _ptr(state), _ptr(u_ref),
                        _ptr(obs), _ptr(obs_type), _ptr(u_out), _ptr(status), _ptr(beta_out), C.c_void_p(stream))
            else:
                fn = self._lib.sccbf_solve_f32 if dt == torch.float32 else self._lib.sccbf_solve_f64
                rc = fn(C.byref(self.params), A, M, _ptr(state), _ptr(u_ref), _ptr(obs), _ptr(obs_type),
                        _ptr(u_out), _ptr(status), _ptr(rows_out), C.c_void_p(stream))
        _lib.check(rc)
        if rows_out is not None:
            return u_out, status, rows_out
        return u_out, status

    def bind_batches(self, table, first=0, count=None):
        """solve_batches(table, first, count) with everything looked up and converted once: returns launch(),
        which enqueues the launch on `stream` (default: the stream current NOW) and costs the host a few
        microseconds less per call.  The filter's parameters are read at bind time."""
        K = len(table) - first if count is None else int(count)
        if first < 0 or K < 0 or first + K > len(table):
            raise ValueError("batches %d ... %d of a table of %d" % (first, first + K, len(table)))
        params = _lib.Params.from_buffer_copy(self.params)
        params.flags = (table.flags | (FLAG_SLIP_IO if self.slip_io else 0)
                        | (_lib.FLAG_OBS_STREAMED if self.streamed_obs else 0))
        fn = self._lib.sccbf_solve_batches_f32 if table.dtype == torch.float32 else self._lib.sccbf_solve_batches_f64
        args = (C.byref(params), C.c_int32(K), C.c_int32(table.M), C.c_int64(table.a_max(first, K)),
                C.c_void_p(table.records.data_ptr() + 64 * first),
                C.c_void_p(torch.cuda.current_stream(table.device).cuda_stream))
        check = _lib.check

        def launch():
            check(fn(*args))
        launch._keep = (params, table)
        return launch

    def solve_batches(self, table, first=0, count=None):
        """One launch for `count` batches of a BatchTable from record `first` on (default: all of them):
        sccbf_solve_batches_*, the answers of solve() called per batch, bit for bit, at the memory rate of
        one launch over all their agents.  Dynamics "dbm" or "kbm".  Asynchronous on the current stream."""
        K = len(table) - first if count is None else int(count)
        if first < 0 or K < 0 or first + K > len(table):
            raise ValueError("batches %d ... %d of a table of %d" % (first, first + K, len(table)))
        self.params.flags = (table.flags | (FLAG_SLIP_IO if self.slip_io else 0)
                             | (_lib.FLAG_OBS_STREAMED if self.streamed_obs else 0))
        fn = self._lib.sccbf_solve_batches_f32 if table.dtype == torch.float32 else self._lib.sccbf_solve_batches_f64
        with torch.cuda.device(table.device):
            stream = torch.cuda.current_stream(table.device).cuda_stream
            rc = fn(C.byref(self.params), K, table.M, table.a_max(first, K),
                    C.c_void_p(table.records.data_ptr() + 64 * first), C.c_void_p(stream))
        _lib.check(rc)

    def solve_batches_static(self, table, first=0, count=None):
        """solve_batches() without a record table in device memory of the caller's: the records go to the
        library's own table on the GPU (sccbf_solve_batches_static; at most _lib.MAX_STATIC_BATCHES batches per
        call, nothing is allocated).  `table` is a BatchTable, of which only the host copy of the records is used.
        Calls on one GPU share the library's table: keep them on one stream, or serialise them."""
        K = len(table) - first if count is None else int(count)
        if first < 0 or K < 0 or first + K > len(table):
            raise ValueError("batches %d ... %d of a table of %d" % (first, first + K, len(table)))
        self.params.flags = (table.flags | (FLAG_SLIP_IO if self.slip_io else 0)
                             | (_lib.FLAG_OBS_STREAMED if self.streamed_obs else 0))
        rec = np.ascontiguousarray(table.host_records[first:first + K])
        with torch.cuda.device(table.device):
            stream = torch.cuda.current_stream(table.device).cuda_stream
            rc = self._lib.sccbf_solve_batches_static(C.byref(self.params), 4 if table.dtype == torch.float32 else 8,
                                                      K, table.M, C.c_void_p(rec.ctypes.data), C.c_void_p(stream))
        _lib.check(rc)

    def prepare_obs(self, obs, obs_type, agents=None, frozen_cone_beta=False):
        """As-written ellipse / cone slots -> their prepared kinds, in place: the module's prepare_obs().
        A prepared cone keeps the beta it was prepared with.  The steer-rate model hands every cone the agent's
        slip angle after each solve (reference cbf/cbf.py:424-427; cone_beta_state=True here), which a prepared
        cone cannot follow: with dynamics "sadbm" and collision cones among the slots this raises unless
        frozen_cone_beta=True says that the frozen value is wanted."""
        if (self.params.dynamics == _lib.DYN_SADBM and not frozen_cone_beta
                and bool((obs_type == OBS_CONE).any().item())):
            raise ValueError("prepare_obs with dynamics 'sadbm': prepared collision cones keep the beta they were prepared "
                             "with and do not follow the agent's slip angle (cone_beta_state); keep the cones as written, "
                             "or pass frozen_cone_beta=True")
        return prepare_obs(obs, obs_type, agents)

    def solve_qp(self, rows, u_nom, u_out=None, status=None):
        """QP stage alone: rows (M,A,4) = a0, a1, b, - with a.u >= b; u_nom (A,2)."""
        if not rows.is_cuda:
            raise _lib.SccbfError("BatchedCBF.solve_qp needs tensors on the GPU (no CPU fallback)")
        dt, dev = rows.dtype, rows.device
        M, A = rows.shape[0], rows.shape[1]
        for name, t, shape in (("rows", rows, (M, A, 4)), ("u_nom", u_nom, (A, 2))):
            if tuple(t.shape) != shape or t.dtype != dt or t.device != dev or not t.is_contiguous():
                raise ValueError("%s: expected contiguous %s %s on %s" % (name, shape, dt, dev))
        if u_out is None:
            u_out = torch.empty((A, 2), dtype=dt, device=dev)
        if status is None:
            status = torch.empty((A,), dtype=torch.int32, device=dev)
        fn = self._lib.sccbf_qp_f32 if dt == torch.float32 else self._lib.sccbf_qp_f64
        self.params.flags = _lib.FLAG_INFEASIBLE_MIN_VIOLATION if self.min_violation else 0
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = fn(C.byref(self.params), A, M, _ptr(rows), _ptr(u_nom), _ptr(u_out), _ptr(status),
                    C.c_void_p(stream))
        _lib.check(rc)
        return u_out, status


class BatchTable:
    """The batches of BatchedCBF.solve_batches: a device array of sccbf_batch records (include/sccbf.h) and the
    tensors they point to.  batches: dicts with state (A,4), u_ref (A,2), obs, obs_type as solve() takes them,
    and optionally u_out (A,2), status (A,), rows_out (M,A,4) (made here when missing; rows only with
    want_rows).  All batches share dtype, device, M and the layout of obs / obs_type; A may differ.
    Build it once and launch it every tick: the records hold addresses, the tensors may be rewritten in place."""

    def __init__(self, batches, want_rows=False):
        batches = [dict(b) for b in batches]
        if not batches:
            raise ValueError("no batches")
        b0 = batches[0]
        if not b0["state"].is_cuda:
            raise _lib.SccbfError("BatchTable needs tensors on the GPU (no CPU fallback)")
        self.dtype, self.device = b0["state"].dtype, b0["state"].device
        if self.dtype not in (torch.float32, torch.float64):
            raise TypeError("state must be float32 or float64")
        td = b0["obs_type"].dim()
        lflag, self.M, _ = obs_layout(b0["obs"], b0["state"].shape[0])
        self.flags = lflag | (FLAG_TYPE_PER_AGENT if td == 2 else 0)
        dt, dev, M = self.dtype, self.device, self.M
        rec = np.zeros((len(batches), 8), dtype=np.int64)
        for i, b in enumerate(batches):
            A = b["state"].shape[0]
            exp_obs = obs_shape(lflag, A, M)
            b.setdefault("u_out", torch.empty((A, 2), dtype=dt, device=dev))
            b.setdefault("status", torch.empty((A,), dtype=torch.int32, device=dev))
            if want_rows:
                b.setdefault("rows_out", torch.empty((M, A, 4), dtype=dt, device=dev))
            for name, shape, dtype in (("state", (A, 4), dt), ("u_ref", (A, 2), dt), ("obs", exp_obs, dt),
                                       ("obs_type", (M, A) if td == 2 else (M,), torch.uint8),
                                       ("u_out", (A, 2), dt), ("status", (A,), torch.int32),
                                       ("rows_out", (M, A, 4), dt)):
                t = b.get(name)
                if t is None and name == "rows_out":
                    continue
                if (t is None or tuple(t.shape) != shape or t.dtype != dtype or t.device != dev
                        or not t.is_contiguous()):
                    raise ValueError("batch %d, %s: expected contiguous %s %s on %s" % (i, name, shape, dtype, dev))
            rec[i] = [A] + [b[k].data_ptr() for k in ("state", "u_ref", "obs", "obs_type", "u_out", "status")] + [
                b["rows_out"].data_ptr() if b.get("rows_out") is not None else 0]
        self.batches = batches                      # keeps the tensors alive
        self.agents = [int(a) for a in rec[:, 0]]
        self.host_records = rec                     # (K, 8) int64: struct sccbf_batch as the ABI lays it out
        self.records = torch.from_numpy(rec).to(dev)

    def __len__(self):
        return len(self.batches)

    def a_max(self, first=0, count=None):
        part = self.agents[first:] if count is None else self.agents[first:first + count]
        return max(part) if part else 0


def prewarm():
    """Device set-up that the library otherwise does on first use and that may not happen under stream capture
    (sccbf_prewarm): call it once, with the device current, before capturing a first launch with more than 128
    rows per QP or the first solve_batches_static."""
    _lib.check(_lib.load().sccbf_prewarm())


def prepare_obs(obs, obs_type, agents=None):
    """As-written ellipse / cone slots -> prepared slots (kinds 4 and 5), IN PLACE on the GPU
    (sccbf_prepare_obs_*): obs in any of the four layouts, obs_type (M,) or (M, A) uint8 is
    rewritten to the prepared kinds.  agents: A, needed for tiles whose last tile is partly unused
    (default: every lane of every tile) and for per-agent lists of two slots.  Call it when obstacles change; the filter then does no
    sine, cosine or reciprocal per slot and tick.  Returns (obs, obs_type)."""
    if not obs.is_cuda:
        raise _lib.SccbfError("prepare_obs needs tensors on the GPU (no CPU fallback)")
    lib = _lib.load()
    dt, dev = obs.dtype, obs.device
    flags, M, exp = obs_layout(obs, agents)
    if flags == FLAG_OBS_TILED:
        A = obs.shape[0] * 64 if agents is None else int(agents)
    else:
        A = {FLAG_OBS_SHARED: 1, _lib.FLAG_OBS_AGENT_MAJOR: obs.shape[0]}.get(flags, obs.shape[2])
    if obs_type.dim() == 2:
        flags |= FLAG_TYPE_PER_AGENT
        exp_ty = (M, A)
    else:
        exp_ty = (M,)
    if dt not in (torch.float32, torch.float64) or tuple(obs.shape) != exp or not obs.is_contiguous():
        raise ValueError("obs: expected a contiguous float32 / float64 tensor of shape %s" % (exp,))
    if (tuple(obs_type.shape) != exp_ty or obs_type.dtype != torch.uint8 or obs_type.device != dev
            or not obs_type.is_contiguous()):
        raise ValueError("obs_type: expected contiguous uint8 %s on %s" % (exp_ty, dev))
    fn = lib.sccbf_prepare_obs_f32 if dt == torch.float32 else lib.sccbf_prepare_obs_f64
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(fn(A, M, _ptr(obs), _ptr(obs_type), flags, C.c_void_p(stream)))
    return obs, obs_type


# ---------------------------------------------------------------- slot packing helpers
def ellipse_slots(cx, cy, a, b, theta=0.0, vx=0.0, vy=0.0):
    """Rows of 8 scalars for SCCBF_OBS_ELLIPSE (a, b already include any buffer)."""
    return _stack8(cx, cy, a, b, theta, vx, vy, 0.0)


def cone_slots(cx, cy, theta_o, v_o, a, beta=0.0):
    """SCCBF_OBS_CONE slots (a already includes the buffer, reference default 1.5)."""
    return _stack8(cx, cy, theta_o, v_o, a, beta, 0.0, 0.0)


def lane_slots(c0, c1=0.0, c2=0.0, c3=0.0, buffer=1.5):
    return _stack8(c0, c1, c2, c3, buffer, 0.0, 0.0, 0.0)


def lane6_slots(c, buffer=1.5):
    """SCCBF_OBS_LANE6 slots from up to seven coefficients c = (c0, c1, ...), y = c0 + c1 x + ... + c6 x^6."""
    c = list(c) + [0.0] * (7 - len(c))
    return _stack8(*c[:7], buffer)


def radial_slots(cx, cy, a, b, cdot_x=0.0, cdot_y=0.0, kv=1.0):
    return _stack8(cx, cy, a, b, cdot_x, cdot_y, kv, 0.0)


def _stack8(*cols):
    ref = None
    for c in cols:
        if torch.is_tensor(c):
            ref = c
            break
    if ref is None:
        return torch.tensor([float(c) for c in cols], dtype=torch.float64)
    out = [c if torch.is_tensor(c) else torch.full_like(ref, float(c)) for c in cols]
    return torch.stack(out, dim=-1)


def pack_obs_tiled(slots):
    """(M, A, 8) per-agent slots -> tiled ABI layout (ceil(A/64), M, 2, 64, 4); the unused
    lanes of a last partial tile are zero."""
    M, A, _ = slots.shape
    T = (A + 63) // 64
    out = slots.new_zeros((T * 64, M, 2, 4))
    out[:A] = slots.reshape(M, A, 2, 4).permute(1, 0, 2, 3)
    return out.reshape(T, 64, M, 2, 4).permute(0, 2, 3, 1, 4).contiguous()


def obs_shape(layout_flag, A, M):
    """Shape of obs for A agents and M slots in the layout a flag names (0: planes)."""
    return {FLAG_OBS_TILED: ((A + 63) // 64, M, 2, 64, 4), FLAG_OBS_SHARED: (M, 2, 4),
            _lib.FLAG_OBS_AGENT_MAJOR: (A, M, 2, 4)}.get(layout_flag, (M, 2, A, 4))


def obs_layout(obs, A=None):
    """The layout of an obstacle tensor by its shape -> (layout flag, M, the shape it should have for A agents):
    (ceil(A/64),M,2,64,4) tiles, (M,2,4) one shared table, (M,2,A,4) planes, (A,M,2,4) per-agent lists.  A
    four-dimensional tensor that fits both readings (A = M = 2, or two slots with A unknown) is read as
    planes."""
    if not torch.is_tensor(obs) or obs.dim() not in (3, 4, 5):
        raise ValueError("obs: expected planes (M,2,A,4), tiles (ceil(A/64),M,2,64,4), per-agent lists "
                         "(A,M,2,4) or a shared table (M,2,4)")
    if obs.dim() == 5:
        flag, M = FLAG_OBS_TILED, obs.shape[1]
    elif obs.dim() == 3:
        flag, M = FLAG_OBS_SHARED, obs.shape[0]
    elif obs.shape[1] == 2 and (A is None or obs.shape[2] == A):
        flag, M = 0, obs.shape[0]
    elif obs.shape[2] == 2:
        flag, M = _lib.FLAG_OBS_AGENT_MAJOR, obs.shape[1]
    else:
        flag, M = 0, obs.shape[0]
    if A is None:
        A = {FLAG_OBS_TILED: obs.shape[0] * 64, FLAG_OBS_SHARED: 1,
             _lib.FLAG_OBS_AGENT_MAJOR: obs.shape[0]}.get(flag, obs.shape[2])
    return flag, M, obs_shape(flag, int(A), M)


def planes_to_agents(obs):
    """(M, 2, A, 4) -> (A, M, 2, 4): every agent's slots side by side (SCCBF_FLAG_OBS_AGENT_MAJOR), the layout
    that lists of more than 128 slots stream fastest from."""
    return obs.permute(2, 0, 1, 3).contiguous()


def planes_to_tiled(obs):
    """(M, 2, A, 4) -> (ceil(A/64), M, 2, 64, 4)."""
    M, _, A, _ = obs.shape
    T = (A + 63) // 64
    out = obs.new_zeros((M, 2, T * 64, 4))
    out[:, :, :A] = obs
    return out.reshape(M, 2, T, 64, 4).permute(2, 0, 1, 3, 4).contiguous()


def pack_obs(slots):
    """(M, A, 8) per-agent slots (or (M, 8) shared) -> the ABI layout (M,2,A,4) / (M,2,4)."""
    if slots.dim() == 2:
        return slots.reshape(slots.shape[0], 2, 4).contiguous()
    M, A, _ = slots.shape
    return slots.reshape(M, A, 2, 4).permute(0, 2, 1, 3).contiguous()


def capture_independent(launches, branches=3, stream=None):
    """One hipGraph for a list of INDEPENDENT launches (callables that enqueue on the current stream, each with
    its own inputs and outputs: the batches of a Monte-Carlo sweep), dealt to `branches` parallel branches
    of the graph: side streams forked from and joined to the capture stream.  A 65536-agent x 16-row launch
    fills four wavefronts per SIMD once, all of them loading and then all of them computing; with three launches
    in flight one loads while another computes: 5.7 instead of 9.1 us per launch on an MI355X.  Launches that
    depend on each other (the ticks of one closed loop) belong on one branch: `branches=1`.
    Call every launch once eagerly first (warm-up, and kernels that need more than 64 KB of LDS set their
    attribute then).  stream: the capture stream (default: the current stream, or a new one when that is the
    default stream, which cannot capture).  Returns the torch.cuda.CUDAGraph; replay() enqueues it on the stream
    that is current then."""
    launches = list(launches)
    cur = torch.cuda.current_stream()
    if stream is None:
        stream = cur if cur != torch.cuda.default_stream(cur.device) else torch.cuda.Stream(cur.device)
    stream.wait_stream(cur)        # what the caller enqueued so far comes first
    branches = max(1, min(int(branches), len(launches)))
    side = [torch.cuda.Stream(stream.device) for _ in range(branches - 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for s in side:
            s.wait_stream(stream)
        for i, f in enumerate(launches):
            with torch.cuda.stream(([stream] + side)[i % branches]):
                f()
        for s in side:
            stream.wait_stream(s)
    cur.wait_stream(stream)
    g._sccbf_streams = side + [stream]      # the streams live as long as the graph
    return g
