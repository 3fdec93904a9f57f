"""CBF-QP filter classes with the reference's names and call pattern (cbf/cbf.py).

    cbf = DBM_CBF_2DS(alpha=1.0)
    cbf.set_model_params(lr=lr, lf=lf)
    cbf.obstacle_list2d.update({0: CollisionCone2D(a_cone, s, s_obs)})
    cbf.update_state(s=s)
    cbf.set_qp_cost_weight(np.diag([0.5, 0.5]))
    a, delta = cbf.solve_cbf(np.array([a_ref, delta_ref]))

The obstacle rows and the 2-input QP run on the GPU through sccbf_solve_f64; the
reference hands the same problem to cvxopt.solvers.cp (cbf/cbf.py:213).  Results are
length-2 float arrays (the reference returns a 2x1 cvxopt matrix, indexed u[0], u[1]).
"""
import ctypes as C
import time
import warnings

import numpy as np

from .. import _lib
from . import _device
from ._vec import Point2
from .obstacles import Obstacle2DTypes, ObstacleList2D
from .utils import ZERO_TOL

_STATUS = {0: "optimal", 1: "infeasible", 2: "non-finite input"}


class CBFStatusWarning(RuntimeWarning):
    """The filter could not produce a safe input and handed back the nominal one."""


def _note_status(flt, status):
    """Keep the device's verdict on the filter object and say so when the answer is the unfiltered
    nominal input (the reference never looks at cvxopt's status, cbf/cbf.py:213-220)."""
    flt.last_status = int(status)
    if status == 1 and getattr(flt, "infeasible_policy", "nominal") == "min_violation":
        warnings.warn("CBF-QP infeasible: no input satisfies every row; solve_cbf returns the input whose largest row "
                      "violation is least (last_status = 1)", CBFStatusWarning, stacklevel=3)
    elif status != 0:
        warnings.warn("CBF-QP %s: solve_cbf returns the nominal input unchanged (last_status = %d)"
                      % (_STATUS.get(status, "unknown"), status), CBFStatusWarning, stacklevel=3)


class _FilterBase:
    _dynamics = _lib.DYN_DBM

    def __init__(self, alpha=1.0):
        self.obstacle_list2d = ObstacleList2D()
        self._alpha = alpha
        self._p = Point2()
        self._v = 0
        self._theta = 0
        self._R = np.eye(2)
        self._lr = 1.45
        self._lf = 1.45
        self._L = 2.9
        self.s = np.zeros(4)
        self.last_status = 0   # 0 optimal, 1 infeasible, 2 non-finite input: of the latest solve_cbf
        # what solve_cbf returns when no input satisfies every row (the reference applies whatever cvxopt hands back,
        # cbf/cbf.py:213-220): "nominal" = u_ref unchanged, "min_violation" = the input whose largest row violation
        # (distance to the row's half-plane in the metric of R) is least; DBM, KBM and DUM filters
        self.infeasible_policy = "nominal"

    def set_alpha(self, alpha=1.0):
        self._alpha = alpha

    def set_qp_cost_weight(self, R):
        R = np.asarray(R, dtype=np.float64)
        if R.shape != (2, 2):
            raise ValueError("Expected a symmetrix matrix of size 2 as input. Please check the "
                             "value of the matrix R you are using.")
        self._R = R

    def _params(self, flags=0):
        p = _lib.Params()
        _lib.load().sccbf_default_params(C.byref(p))
        p.alpha = float(self._alpha)
        p.lr, p.lf, p.L = float(self._lr), float(self._lf), float(self._L)
        for i in range(4):
            p.R[i] = float(self._R.reshape(-1)[i])
        p.dynamics = self._dynamics
        p.flags = flags
        return p

    def _solve(self, u_ref):
        if len(self.obstacle_list2d) < 1:
            raise ValueError("Cannot solve CBF for an empty obstacle list. Update the obstacle "
                             "list so that it is non-empty in order to move forward.")
        u_ref = np.asarray(u_ref, dtype=np.float64).reshape(-1)
        kinds, slots = self.obstacle_list2d.slots()
        if self.infeasible_policy not in ("nominal", "min_violation"):
            raise ValueError("infeasible_policy: 'nominal' or 'min_violation'")
        flags = _lib.FLAG_INFEASIBLE_MIN_VIOLATION if self.infeasible_policy == "min_violation" else 0
        u, status, rows = _device.solve(self._params(flags), self.s, u_ref, kinds, slots)
        _note_status(self, status)
        info = {"x": u.copy(), "status": _STATUS.get(status, "unknown"), "status_code": status,
                "rows": rows}
        return info, u


class DBM_CBF_2DS(_FilterBase):
    """Acceleration-controlled dynamic bicycle model, u = [a, delta] (reference
    cbf/cbf.py:112-220).  Supports Ellipse2D, CollisionCone2D, PolyLane, RadialEllipse2D."""
    _dynamics = _lib.DYN_DBM

    def update_state(self, s, s_obs_dict=None, buffer=None, **kwargs):
        self.s = np.asarray(s, dtype=np.float64).reshape(-1)
        self.s_obs_dict = s_obs_dict
        self._p = Point2(self.s[0], self.s[1])
        self._theta = self.s[2]
        self._v = self.s[3]
        self.obstacle_list2d.update_state(s=self.s, s_obs_dict=s_obs_dict, buffer=buffer)

    def set_model_params(self, lr, lf):
        self._lr = lr
        self._lf = lf

    def gc(self, *args, **kwargs):
        """Input matrix g(x) of the model, 4x2 (reference cbf/cbf.py:159-161)."""
        v, th = self._v, self._theta
        return np.array([[0.0, -v * np.sin(th)], [0.0, v * np.cos(th)], [0.0, v / self._lr], [1.0, 0.0]])

    def fc(self, *args, **kwargs):
        """Drift f(x), 4x1 (reference cbf/cbf.py:163-164)."""
        v, th = self._v, self._theta
        return np.array([[v * np.cos(th)], [v * np.sin(th)], [0.0], [0.0]])

    def solve_cbf(self, u_ref, return_solver=False):
        info, u = self._solve(u_ref)
        if return_solver:
            return info, u
        return u


class DUM_CBF_2DS(DBM_CBF_2DS):
    """Dynamic unicycle model, u = [a, omega] (reference cbf/cbf.py:222-298): the acceleration
    drives v, the second input drives the yaw directly, no steering transform.  The reference
    declares its drift 5x1 with four entries (243-245) and does not run as shipped; this is the
    model with the 4x1 drift."""
    _dynamics = _lib.DYN_DUM

    def gc(self, *args, **kwargs):
        """4x2 (reference cbf/cbf.py:237-239, column lists [0,0,0,1] and [0,0,1,0])."""
        return np.array([[0.0, 0.0], [0.0, 0.0], [0.0, 1.0], [1.0, 0.0]])

    def fc(self, *args, **kwargs):
        v, th = self._v, self._theta
        return np.array([[v * np.cos(th)], [v * np.sin(th)], [0.0], [0.0]])


class SADBM_CBF_2DS(DBM_CBF_2DS):
    """State-augmented, steer-rate controlled dynamic bicycle model (reference
    cbf/cbf.py:301-437): the slip angle beta is a fifth state, the QP works on
    [a, d(beta)/dt], solve_cbf takes and returns [a, delta].  dt=None measures the time
    between calls (cbf.py:361-367, 416-417)."""
    _dynamics = _lib.DYN_SADBM

    def __init__(self, alpha=1.0, dt=0.001):
        super().__init__(alpha)
        self._DT_MODE_AUTO = 0
        if dt is None:
            self._DT_MODE_AUTO = 1
            self._dt = 1e-6
        else:
            self._dt = dt
        self.t_last = time.time()
        self.beta_ref_last = 0.0
        self.beta_ref_dot = 0.0
        self._beta = 0.0

    def gc(self, *args, **kwargs):
        """5x2 (reference cbf/cbf.py:338-340)."""
        return np.array([[0.0, 0.0], [0.0, 0.0], [0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])

    def fc(self, *args, **kwargs):
        """5x1 (reference cbf/cbf.py:342-347)."""
        v, th, b = self._v, self._theta, self._beta
        return np.array([[v * np.cos(th + b)], [v * np.sin(th + b)], [v * np.sin(b) / self._lr], [0.0], [0.0]])

    def solve_cbf(self, u_ref, return_solver=False):
        u_ref = np.asarray(u_ref, dtype=np.float64).reshape(-1)
        self.t_current = time.time()
        if self._DT_MODE_AUTO:
            self._dt = max(self.t_current - self.t_last, ZERO_TOL)
        if len(self.obstacle_list2d) < 1:
            # the reference has already advanced its nominal slip angle when it raises (369, 375-378)
            beta_ref = np.arctan2(self._lr * np.tan(u_ref[1]), self._lf + self._lr)
            self.beta_ref_dot = (beta_ref - self.beta_ref_last) / self._dt
            self.beta_ref_last = beta_ref
            raise ValueError("Cannot solve CBF for an empty obstacle list. Update the obstacle "
                             "list so that it is non-empty in order to move forward.")
        if self.infeasible_policy != "nominal":
            raise ValueError("the steer-rate model has advanced its slip state with the nominal input by the time the rows "
                             "turn out to admit none: infeasible_policy 'nominal' only")
        kinds, slots = self.obstacle_list2d.slots()
        last = self.beta_ref_last
        u, status, rows, slip = _device.solve_sa(self._params(), self._dt, self.s, [self._beta, last], u_ref,
                                                 kinds, slots)
        self._beta, self.beta_ref_last = float(slip[0]), float(slip[1])
        _note_status(self, status)
        self.beta_ref_dot = (self.beta_ref_last - last) / self._dt
        # every collision cone follows the filter's slip angle (cbf.py:424-427)
        for obstacle in self.obstacle_list2d.values():
            if obstacle.type == Obstacle2DTypes.COLLISION_CONE2D:
                obstacle.update(beta=self._beta)
        self.t_last = time.time()
        info = {"x": u.copy(), "status": _STATUS.get(status, "unknown"), "status_code": status, "rows": rows}
        if return_solver:
            return info, u
        return u


class KBM_VC_CBF2D(_FilterBase):
    """Velocity-controlled kinematic bicycle model, u = [v, delta] (reference
    cbf/cbf.py:33-110); rows use position and heading only."""
    _dynamics = _lib.DYN_KBM

    def update_state(self, p, theta):
        self._p = p
        self._theta = theta
        self.s = np.array([p.x, p.y, theta, 0.0])
        self.obstacle_list2d.update_state(s=self.s)

    def set_model_params(self, L):
        self._L = L

    def solve_cbf(self, u_ref):
        info, u = self._solve(u_ref)
        return info, u
