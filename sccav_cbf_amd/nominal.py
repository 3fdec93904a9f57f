"""Batched nominal controller, vehicle step and closed-loop rollout around the filter.

BatchedNominal  Stanley steering + PID speed control for every agent of a batch on one shared
                course (the per-ego objects of reference cbf/controllers.py:17-180; the driver
                scripts restate the law in test_scripts/stanley_controller_ellipse.py:146-212).
vehicle_step    State.update_com of the drivers (stanley_controller_ellipse.py:122-131).
ClosedLoop      nominal -> filter -> vehicle step, tick after tick, without leaving the GPU
                (the main loop of stanley_controller_ellipse.py:700-760 for a whole batch).

All arithmetic is in libsccbf.so; tensors hold device memory and the current stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .batched import BatchedCBF, _ptr, obs_layout


def _check(name, t, shape, dtype, dev):
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != dev or not t.is_contiguous():
        raise ValueError("%s: expected contiguous %s %s on %s, got %s %s on %s" % (
            name, tuple(shape), dtype, dev, tuple(t.shape), t.dtype, t.device))


class BatchedNominal:
    """u_ref = (a_ref, delta_ref) for A agents following one course.

    axle is the distance from the state's reference point to the front axle: lf for
    LateralStanley (cbf/controllers.py:80-81), the wheel base for the driver scripts."""

    def __init__(self, axle=2.0, k=0.5, ks=0.01, kp=1.0, ki=0.0, kd=0.0, dt=0.1,
                 target_speed=30.0 / 3.6, waypoint_speed=False):
        self._lib = _lib.load()
        self.params = _lib.NominalParams()
        self._lib.sccbf_default_nominal_params(C.byref(self.params))
        self.params.axle, self.params.k, self.params.ks = float(axle), float(k), float(ks)
        self.params.kp, self.params.ki, self.params.kd = float(kp), float(ki), float(kd)
        self.params.dt = float(dt)
        self.params.target_speed = float(target_speed)
        self.params.flags = _lib.NOMINAL_WAYPOINT_SPEED if waypoint_speed else 0

    def set_gains(self, k=None, ks=None, kp=None, ki=None, kd=None):
        for name, v in (("k", k), ("ks", ks), ("kp", kp), ("ki", ki), ("kd", kd)):
            if v is not None:
                setattr(self.params, name, float(v))

    def control(self, state, traj, target_idx, pid_state=None, u_ref=None):
        """state (A,4), traj (N,4) = x, y, yaw, v_des, target_idx (A,) int32 (updated in place),
        pid_state (A,2) = previous error, integral (updated in place; None = P control).
        Returns u_ref (A,2).  Asynchronous on the current stream."""
        if not state.is_cuda:
            raise _lib.SccbfError("BatchedNominal.control needs tensors on the GPU (no CPU fallback)")
        dt, dev = state.dtype, state.device
        if dt not in (torch.float32, torch.float64):
            raise ValueError("state must be float32 or float64")
        A, N = state.shape[0], traj.shape[0]
        if N < 1:
            raise ValueError("the course needs at least one way-point")
        _check("state", state, (A, 4), dt, dev)
        _check("traj", traj, (N, 4), dt, dev)
        _check("target_idx", target_idx, (A,), torch.int32, dev)
        if pid_state is not None:
            _check("pid_state", pid_state, (A, 2), dt, dev)
        if u_ref is None:
            u_ref = torch.empty((A, 2), dtype=dt, device=dev)
        _check("u_ref", u_ref, (A, 2), dt, dev)
        fn = self._lib.sccbf_nominal_f32 if dt == torch.float32 else self._lib.sccbf_nominal_f64
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = fn(C.byref(self.params), A, N, _ptr(state), _ptr(traj), _ptr(target_idx), _ptr(pid_state),
                    _ptr(u_ref), C.c_void_p(stream))
        _lib.check(rc)
        return u_ref


def vehicle_step(state, u, lr=1.45, lf=1.45, max_steer=np.radians(30.0), dt=0.1, beta_out=None):
    """In-place bicycle step of state (A,4) with u (A,2) = a, delta; beta_out (A,) optional."""
    if not state.is_cuda:
        raise _lib.SccbfError("vehicle_step needs tensors on the GPU (no CPU fallback)")
    ty, dev = state.dtype, state.device
    A = state.shape[0]
    _check("state", state, (A, 4), ty, dev)
    _check("u", u, (A, 2), ty, dev)
    if beta_out is not None:
        _check("beta_out", beta_out, (A,), ty, dev)
    lib = _lib.load()
    fn = lib.sccbf_vehicle_step_f32 if ty == torch.float32 else lib.sccbf_vehicle_step_f64
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = fn(A, _ptr(state), _ptr(u), float(lr), float(lf), float(max_steer), float(dt), _ptr(beta_out),
                C.c_void_p(stream))
    _lib.check(rc)
    return state


def vehicle_step_kin(state, u, mode, L=2.9, max_steer=np.radians(30.0), dt=0.1):
    """In-place kinematic step of state (A,4) (sccbf_vehicle_step_kin_*): mode "acc" = State.update
    with u = (a, delta), "vel" = State.update_by_vel with u = (v, delta) (reference
    test_scripts/stanley_controller_ellipse.py:86-118), "unicycle" = u = (a, omega)."""
    if not state.is_cuda:
        raise _lib.SccbfError("vehicle_step_kin needs tensors on the GPU (no CPU fallback)")
    ty, dev = state.dtype, state.device
    A = state.shape[0]
    _check("state", state, (A, 4), ty, dev)
    _check("u", u, (A, 2), ty, dev)
    code = {"acc": _lib.STEP_KIN_ACC, "vel": _lib.STEP_KIN_VEL, "unicycle": _lib.STEP_UNICYCLE}[mode]
    lib = _lib.load()
    fn = lib.sccbf_vehicle_step_kin_f32 if ty == torch.float32 else lib.sccbf_vehicle_step_kin_f64
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(fn(A, _ptr(state), _ptr(u), code, float(L), float(max_steer), float(dt), C.c_void_p(stream)))
    return state


class ClosedLoop:
    """nominal -> filter -> vehicle step for a batch, every tensor resident on the GPU.

    tick() launches two or three kernels on the current stream and, for the "dbm" and "sadbm" models,
    allocates nothing, so a caller may capture a number of ticks in a hipGraph ("kbm" and "dum" convert
    the nominal input with torch element-wise ops, whose temporaries come from torch's allocator: capturable
    too, through torch's graph pool).  Obstacles are whatever `obs` holds when the tick
    runs: move them between ticks (or inside the captured region) with ordinary tensor ops.

    nominal=None, traj=None: no course, the nominal input is (0, 0) on every tick (the stationary
    ego of test_scripts/radial_dynamic_obstacles.py:444).  The vehicle model follows the filter's:
    "dbm" (a, delta) -> update_com; "kbm" (v, delta), v the tracked speed -> update_by_vel; "dum"
    (a, omega = v tan(delta_ref) / L) -> unicycle step; "sadbm": the slip state lives in self.slip
    (beta, beta_ref_last), every cone follows the filter's beta, as reference cbf/cbf.py:424-427 leaves
    them, and the step is update_com with the filter's (a, delta).  tick() and rollout() agree."""

    def __init__(self, cbf: BatchedCBF, nominal: BatchedNominal, traj, state, obs, obs_type,
                 max_steer=np.radians(30.0), dt=0.1, use_pid_state=False):
        self.cbf, self.nominal = cbf, nominal
        self.traj, self.state, self.obs, self.obs_type = traj, state, obs, obs_type
        self.max_steer, self.dt = float(max_steer), float(dt)
        A, dev, ty = state.shape[0], state.device, state.dtype
        if (nominal is None) != (traj is None):
            raise ValueError("give both a nominal controller and a course, or neither")
        self._check_inputs()
        self.target_idx = torch.zeros((A,), dtype=torch.int32, device=dev)
        self.pid_state = torch.zeros((A, 2), dtype=ty, device=dev) if use_pid_state else None
        self.u_ref = torch.zeros((A, 2), dtype=ty, device=dev)
        self.u = torch.empty((A, 2), dtype=ty, device=dev)
        self.status = torch.empty((A,), dtype=torch.int32, device=dev)
        self.beta = torch.empty((A,), dtype=ty, device=dev)
        # steer-rate model (SADBM_CBF_2DS): the filter integrates its own slip angle; the cones follow it
        self.slip = torch.zeros((A, 2), dtype=ty, device=dev) if cbf.params.dynamics == _lib.DYN_SADBM else None
        if self.slip is not None and bool((obs_type == _lib.OBS_CONE_PREP).any().item()):
            # every cone of this model follows the agent's slip angle (reference cbf/cbf.py:424-427); a prepared cone
            # holds the velocity resolved with the beta it was prepared with and would silently stay there
            raise ValueError("ClosedLoop with the steer-rate model: prepared collision cones (kind 5) cannot follow the "
                             "agent's slip angle; keep the cones as written (kind 1)")

    def _check_inputs(self):
        """Shapes, dtypes, device and contiguity of every tensor the kernels index: a tensor built
        for another batch size, precision or layout must raise here, never reach a launch.
        Returns (M, layout flags)."""
        state = self.state
        if not torch.is_tensor(state) or state.dim() != 2:
            raise ValueError("state: expected a (A, 4) tensor")
        A, dev, ty = state.shape[0], state.device, state.dtype
        if ty not in (torch.float32, torch.float64):
            raise ValueError("state must be float32 or float64")
        _check("state", state, (A, 4), ty, dev)
        obs = self.obs
        flags, M, exp = obs_layout(obs, A)
        _check("obs", obs, exp, ty, dev)
        if not torch.is_tensor(self.obs_type) or self.obs_type.dim() not in (1, 2):
            raise ValueError("obs_type: expected uint8 (M,) or (M, A)")
        if self.obs_type.dim() == 2:
            flags |= _lib.FLAG_TYPE_PER_AGENT
            _check("obs_type", self.obs_type, (M, A), torch.uint8, dev)
        else:
            _check("obs_type", self.obs_type, (M,), torch.uint8, dev)
        if self.traj is not None:
            if not torch.is_tensor(self.traj) or self.traj.dim() != 2 or self.traj.shape[0] < 1:
                raise ValueError("traj: expected a (N, 4) tensor with at least one way-point")
            _check("traj", self.traj, (self.traj.shape[0], 4), ty, dev)
        for name, t, shape, dt in (("target_idx", getattr(self, "target_idx", None), (A,), torch.int32),
                                   ("pid_state", getattr(self, "pid_state", None), (A, 2), ty),
                                   ("slip", getattr(self, "slip", None), (A, 2), ty)):
            if t is not None:
                _check(name, t, shape, dt, dev)
        return M, flags

    def rollout(self, ticks, motion="static", trace=False, seek_gain=0.2, seek_v_min=3.0, out=None):
        """`ticks` ticks in ONE launch (sccbf_rollout_*): state, target_idx and pid_state advance in
        place; obstacles may move inside the launch, with constant velocity or as seekers of
        radial_dynamic_obstacles.py:193-239 (`obs` itself is not written).
        Returns a dict: h_min (A,), n_active (A,), n_infeasible (A,) and, if asked for,
        trace (ticks, A, 4) = a*, delta*, beta, status per tick.  Pass the dict of an earlier call
        as `out` to write into its tensors again: the call then allocates nothing and may be
        captured in a hipGraph."""
        A, dev, ty = self.state.shape[0], self.state.device, self.state.dtype
        obs = self.obs
        M, flags = self._check_inputs()
        if flags & _lib.FLAG_OBS_SHARED:
            raise ValueError("rollout needs per-agent obstacle slots (planes or tiles)")
        if self.cbf.slip_io:
            raise ValueError("rollout works in the caller's coordinates (steering angles): slip_io must be off")
        if int(ticks) < 0:
            raise ValueError("ticks must not be negative")
        if not self.state.is_cuda:
            raise _lib.SccbfError("ClosedLoop.rollout needs tensors on the GPU (no CPU fallback)")
        N = 0 if self.traj is None else self.traj.shape[0]
        rp = _lib.RolloutParams()
        lib = _lib.load()
        lib.sccbf_default_rollout_params(C.byref(rp))
        rp.max_steer, rp.dt, rp.ticks = self.max_steer, self.dt, int(ticks)
        rp.motion = {"static": _lib.MOTION_STATIC, "constant_velocity": _lib.MOTION_CONSTANT_VELOCITY,
                     "seeker": _lib.MOTION_SEEKER}[motion]
        rp.seek_gain, rp.seek_v_min = float(seek_gain), float(seek_v_min)
        nominal_params = self.nominal.params if self.nominal is not None else _lib.NominalParams()
        if out is None:
            out = dict(h_min=torch.empty((A,), dtype=ty, device=dev),
                       n_active=torch.empty((A,), dtype=torch.int32, device=dev),
                       n_infeasible=torch.empty((A,), dtype=torch.int32, device=dev))
            if trace:
                out["trace"] = torch.empty((int(ticks), A, 4), dtype=ty, device=dev)
        else:
            _check("out['h_min']", out["h_min"], (A,), ty, dev)
            _check("out['n_active']", out["n_active"], (A,), torch.int32, dev)
            _check("out['n_infeasible']", out["n_infeasible"], (A,), torch.int32, dev)
            if "trace" in out:
                _check("out['trace']", out["trace"], (int(ticks), A, 4), ty, dev)
        params = _lib.Params.from_buffer_copy(self.cbf.params)   # the filter object keeps its own flags
        params.flags = flags
        fn = lib.sccbf_rollout_f32 if ty == torch.float32 else lib.sccbf_rollout_f64
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = fn(C.byref(params), C.byref(nominal_params), C.byref(rp), A, M, N,
                    _ptr(self.state), _ptr(self.traj), _ptr(self.target_idx), _ptr(self.pid_state),
                    _ptr(self.slip), _ptr(obs),
                    _ptr(self.obs_type), _ptr(out["h_min"]), _ptr(out["n_active"]), _ptr(out["n_infeasible"]),
                    _ptr(out.get("trace")), C.c_void_p(stream))
        _lib.check(rc)
        return out

    def tick(self):
        dyn = self.cbf.params.dynamics
        if self.cbf.slip_io:
            raise ValueError("ClosedLoop works in the caller's coordinates (steering angles): slip_io must be off")
        self._check_inputs()
        if self.nominal is not None:
            self.nominal.control(self.state, self.traj, self.target_idx, self.pid_state, u_ref=self.u_ref)
        if dyn == _lib.DYN_SADBM:
            self.cbf.solve(self.state, self.u_ref, self.obs, self.obs_type, u_out=self.u, status=self.status,
                           slip=self.slip, dt=self.dt, cone_beta_state=True)
        elif dyn == _lib.DYN_DBM:
            # filter and vehicle step in one launch
            self.cbf.solve(self.state, self.u_ref, self.obs, self.obs_type, u_out=self.u, status=self.status,
                           step_dt=self.dt, max_steer=self.max_steer, beta_out=self.beta)
            return
        elif dyn == _lib.DYN_KBM:
            # u_ref = (tracked speed, delta_ref): the drivers hand KBM_VC_CBF2D the target speed itself
            # (stanley_controller_ellipse.py:633, 711), then update_by_vel (102-118)
            if self.nominal is not None:
                wps = self.nominal.params.flags & _lib.NOMINAL_WAYPOINT_SPEED
                self.u_ref[:, 0] = (self.traj[self.target_idx.long().clamp(max=self.traj.shape[0] - 1), 3]
                                    if wps else self.nominal.params.target_speed)
            self.cbf.solve(self.state, self.u_ref, self.obs, self.obs_type, u_out=self.u, status=self.status)
            vehicle_step_kin(self.state, self.u, "vel", L=self.cbf.params.L, max_steer=self.max_steer, dt=self.dt)
            return
        else:  # DUM: the yaw rate of the nominal steering, then the unicycle step
            self.u_ref[:, 1] = self.state[:, 3] * torch.tan(self.u_ref[:, 1]) / self.cbf.params.L
            self.cbf.solve(self.state, self.u_ref, self.obs, self.obs_type, u_out=self.u, status=self.status)
            vehicle_step_kin(self.state, self.u, "unicycle", dt=self.dt)
            return
        vehicle_step(self.state, self.u, lr=self.cbf.params.lr, lf=self.cbf.params.lf,
                     max_steer=self.max_steer, dt=self.dt, beta_out=self.beta)
