"""Nominal controllers that produce u_ref for the filter (reference cbf/controllers.py).

LateralStanley: steering = heading error + atan2(k e, v + ks), with e the cross-track
error of the front axle against the nearest trajectory point and a target index that
never moves backwards.  PID1: scalar PID.  Both are per-ego host code, as in the
reference; they sit in front of the GPU path, not on it.
"""
import numpy as np

from ._vec import is_vec2
from .utils import normalize_angle


class LateralStanley:
    def __init__(self, lr=2.0, lf=2.0, k=0.5, ks=0.01):
        self._lr, self._lf = lr, lf
        self._k, self._ks = k, ks
        self._last_target_idx = 0
        self._current_target_idx = 0
        self._x = self._y = self._yaw = self._v = 0.0
        self._traj = None

    def update_state(self, x, y, yaw, v):
        self._x, self._y, self._yaw, self._v = x, y, yaw, v

    def set_gains(self, k, ks):
        self._k, self._ks = k, ks

    def set_trajectory(self, trajectory):
        """trajectory: sequence of (x, y, yaw, v) way-points."""
        self._traj = np.asarray(trajectory, dtype=np.float64)

    def _calc_target_index(self, front_coords=None):
        if front_coords is None:
            fx = self._x + self._lf * np.cos(self._yaw)
            fy = self._y + self._lf * np.sin(self._yaw)
        else:
            fx, fy = front_coords.x, front_coords.y
        self.fx, self.fy = fx, fy
        dx = fx - self._traj[:, 0]
        dy = fy - self._traj[:, 1]
        idx = int(np.argmin(np.hypot(dx, dy)))
        # error projected on the axle direction (pointing to the vehicle's right)
        ex, ey = -np.cos(self._yaw + np.pi / 2), -np.sin(self._yaw + np.pi / 2)
        return idx, float(dx[idx] * ex + dy[idx] * ey)

    def control(self, trajectory=None, front_coords=None, initial_yaw=0):
        """Returns (delta, target index) (reference cbf/controllers.py:104-151)."""
        if front_coords is not None and not is_vec2(front_coords):
            raise TypeError("Coordinates of the front wheel must be specified as an "
                            "euclid.Vector2() object.")
        if trajectory is not None:
            self.set_trajectory(trajectory)
        idx, err = self._calc_target_index(front_coords)
        if self._last_target_idx >= idx:
            idx = self._last_target_idx
        self._current_target_idx = idx
        theta_e = normalize_angle(self._traj[idx, 2] - self._yaw)
        theta_d = np.arctan2(self._k * err, self._v + self._ks)
        self._last_target_idx = idx
        return float(theta_e + theta_d), idx


class PID1:
    def __init__(self, kp=1.0, kd=0.0, ki=0.0):
        self._kp, self._kd, self._ki = kp, kd, ki
        self._eprev = 0.0
        self._ie = 0.0
        self._dt = 0.1

    def set_gains(self, kp, kd, ki):
        self._kp, self._kd, self._ki = kp, kd, ki

    def set_dt(self, dt):
        self._dt = dt

    def control(self, x, xref):
        """u = kp e + ki int(e) + kd de/dt with e = xref - x (reference 173-180)."""
        e = xref - x
        de = (e - self._eprev) / self._dt
        self._ie += self._dt * e
        self._eprev = e
        return self._kp * e + self._ki * self._ie + self._kd * de
