"""Obstacle objects with the reference's class and method names (cbf/obstacles.py).

Each object is a small parameter record; all barrier arithmetic runs on the GPU:
`slot()` packs the record into the 8-scalar obstacle slot of the C ABI, and the
per-obstacle `f()/dx()/dy()/dtheta()/dv()/dt()/dbeta()` as well as the stacked
versions on `ObstacleList2D` evaluate through `sccbf_partials_f64`.

Differences from the reference that are deliberate (the reference code paths in
question raise or corrupt state, so there is no behaviour to keep):
  * an obstacle's own orientation / velocity are never overwritten by the ego
    state in `update(s=...)` (reference cbf/obstacles.py:253-256 does that to Ellipse2D);
  * `Ellipse2D.update(b=...)` sets b (the reference writes it into a, line 243-244);
  * `ObstacleList2D.update_state` with a dict updates the named obstacles
    (the reference calls dict.update on the container, line 869-872) and raises
    ValueError for a non-dict (the reference builds the exception without raising it).
"""
import enum
import warnings
from collections.abc import MutableMapping

import numpy as np

from . import _device
from ._vec import Point2, Vector2, Vector3, is_vec2
from .geometry import Rotation
from .utils import ZERO_TOL  # noqa: F401  (re-exported like the reference module)

OBS_ELLIPSE, OBS_CONE, OBS_LANE, OBS_RADIAL = 0, 1, 2, 3
OBS_LANE6 = 8


class Obstacle2DTypes(enum.Enum):
    """Reference cbf/obstacles.py:51-57; RADIAL is the time-varying ellipse of
    test_scripts/radial_dynamic_obstacles.py:366-425 as a class of its own."""
    ELLIPSE2D = 0
    COLLISION_CONE2D = 1
    POLY_LANE = 2
    RADIAL_ELLIPSE2D = 3


class BoundingBox:
    """Reference cbf/obstacles.py:59-88 (extent = half sizes, location, rotation, speed)."""

    def __init__(self, extent=None, location=None, rotation=None, velocity=0.0):
        self.extent = extent if extent is not None else Vector3()
        self.location = location if location is not None else Vector3()
        self.rotation = rotation if rotation is not None else Rotation()
        self.velocity = velocity

    def __eq__(self, other):
        return self.location == other.location and self.extent == other.extent

    def __ne__(self, other):
        return not self.__eq__(other)


def _state4(s):
    a = np.asarray(s, dtype=np.float64).reshape(-1)
    if a.size != 4:
        raise ValueError("expected a state vector [x, y, yaw, v], got %d entries" % a.size)
    return a


class Obstacle2DBase:
    """Interface of every 2-D obstacle (reference cbf/obstacles.py:90-137)."""
    type = None

    def __init__(self):
        self.s = np.zeros(4)

    # -- to be provided by subclasses
    def slot(self):
        """(kind, 8 scalars) in the layout of include/sccbf.h."""
        raise NotImplementedError

    # -- evaluation on the GPU
    def _partials(self):
        kind, sl = self.slot()
        return _device.partials(self.s, [kind], np.asarray(sl, dtype=np.float64)[None, :])[0]

    def evaluate(self, *args, **kwargs):
        return float(self._partials()[0])

    def f(self, *args, **kwargs):
        return self.evaluate()

    def dx(self, *args, **kwargs):
        return float(self._partials()[1])

    def dy(self, *args, **kwargs):
        return float(self._partials()[2])

    def dtheta(self, *args, **kwargs):
        return float(self._partials()[3])

    def dv(self, *args, **kwargs):
        return float(self._partials()[4])

    def dt(self, *args, **kwargs):
        return float(self._partials()[5])

    def dbeta(self, *args, **kwargs):
        return 0.0

    def gradient(self, *args, **kwargs):
        p = self._partials()
        return np.array([p[1], p[2], p[3], p[4]])

    def update(self, *args, **kwargs):
        pass

    def update_coords(self, *args, **kwargs):
        pass

    def update_orientation(self, *args, **kwargs):
        pass

    def update_state(self, s, s_obs=None, **kwargs):
        self.update(s=s, s_obs=s_obs)


class _Buffered:
    """apply_buffer / remove_buffer bookkeeping shared by Ellipse2D and CollisionCone2D
    (reference cbf/obstacles.py:167-181, 387-399)."""

    def _grow(self, d):
        raise NotImplementedError

    def apply_buffer(self):
        if not self.BUFFER_FLAG:
            self._grow(self.buffer)
            self.BUFFER_FLAG = True
        else:
            warnings.warn("Warning: Buffer already applied. Call Ignored.")

    def remove_buffer(self):
        if self.BUFFER_FLAG:
            self._grow(-self.buffer)
            self.BUFFER_FLAG = False
        else:
            warnings.warn("Warning: Buffer already removed. Call Ignored.")

    def _set_buffer(self, buffer):
        if self.BUFFER_FLAG:
            self._grow(buffer - self.buffer)
        self.buffer = buffer


class Ellipse2D(Obstacle2DBase, _Buffered):
    """h = (p/a)^2 + (q/b)^2 - 1 in the ellipse frame (reference cbf/obstacles.py:139-331)."""
    type = Obstacle2DTypes.ELLIPSE2D

    def __init__(self, a, b, center=None, theta=0, buffer=0, **kwargs):
        super().__init__()
        if "id" in kwargs:
            self.id = kwargs["id"]
        if center is None:
            center = Vector2(0, 0)
        if not is_vec2(center):
            raise TypeError("Expected an object of type euclid.Vector2 for arg center, but got "
                            + type(center).__name__ + ".")
        self.center = center
        self.theta = theta
        self.vel = Vector2()
        self.a = a + buffer
        self.b = b + buffer
        self.buffer = buffer
        self.BUFFER_FLAG = True

    def __repr__(self):
        return (f"{type(self).__name__}(a = {self.a}, b = {self.b}, center = {self.center}, "
                f"theta = {self.theta}, buffer = {self.buffer}, buffer_applied: {self.BUFFER_FLAG} )\n")

    def _grow(self, d):
        self.a += d
        self.b += d

    def slot(self):
        return OBS_ELLIPSE, [self.center.x, self.center.y, self.a, self.b, self.theta,
                             self.vel.x, self.vel.y, 0.0]

    def update(self, s=None, s_obs=None, center=None, buffer=None, **kwargs):
        if "a" in kwargs:
            self.a = kwargs["a"]
        if "b" in kwargs:
            self.b = kwargs["b"]
        if "theta" in kwargs:
            self.theta = kwargs["theta"]
        if center is not None:
            self.center = center
        if s_obs is not None:
            self.center = Point2(s_obs[0], s_obs[1])
        if s is not None:
            self.s = _state4(s)
        if buffer is not None:
            self._set_buffer(buffer)

    def update_coords(self, xy):
        self.center = xy

    def update_velocity_by_magnitude(self, v):
        """Velocity of magnitude v along the ellipse heading theta (reference 272-277)."""
        self.vel = Vector2(v * np.cos(self.theta), v * np.sin(self.theta))

    def update_velocity(self, v):
        self.vel = v.copy()

    def update_orientation(self, yaw):
        mag = self.vel.magnitude()
        self.theta = yaw
        self.update_velocity_by_magnitude(mag)

    def update_by_bounding_box(self, bbox):
        if not isinstance(bbox, BoundingBox):
            raise TypeError("Expected an object of type cbf.obstacles.BoundingBox, but got "
                            + type(bbox).__name__)
        self.update(a=bbox.extent.x + (self.buffer if self.BUFFER_FLAG else 0.0),
                    b=bbox.extent.y + (self.buffer if self.BUFFER_FLAG else 0.0),
                    center=Vector2(bbox.location.x, bbox.location.y), theta=bbox.rotation.yaw)

    @classmethod
    def from_bounding_box(cls, bbox=None, buffer=0.5, **kwargs):
        if not isinstance(bbox, BoundingBox):
            raise TypeError("Expected an object of type cbf.obstacles.BoundingBox, but got "
                            + type(bbox).__name__)
        return cls(bbox.extent.x, bbox.extent.y, Vector2(bbox.location.x, bbox.location.y),
                   bbox.rotation.yaw, buffer, **kwargs)


class RadialEllipse2D(Ellipse2D):
    """Axis-aligned time-varying ellipse with the speed-dependent margin kv v/(1+v)
    (test_scripts/radial_dynamic_obstacles.py:391-405); `vel` is the centre velocity."""
    type = Obstacle2DTypes.RADIAL_ELLIPSE2D

    def __init__(self, a, b, center=None, kv=1.0, buffer=0, **kwargs):
        super().__init__(a, b, center=center, theta=0, buffer=buffer, **kwargs)
        self.kv = kv

    def slot(self):
        return OBS_RADIAL, [self.center.x, self.center.y, self.a, self.b, self.vel.x, self.vel.y,
                            self.kv, 0.0]


class CollisionCone2D(Obstacle2DBase, _Buffered):
    """Collision-cone barrier for moving obstacles (reference cbf/obstacles.py:333-543):
    h = p_rel . v_rel + |p_rel| |v_rel| cos(phi).  s, s_obs = [x, y, heading, speed]."""
    type = Obstacle2DTypes.COLLISION_CONE2D

    def __init__(self, a=0.0, s=None, s_obs=None, buffer=1.50, **kwargs):
        super().__init__()
        if "id" in kwargs:
            self.id = kwargs["id"]
        self.beta = kwargs.get("beta", 0.0)
        self.s = _state4(s if s is not None else np.zeros(4))
        self.s_obs = _state4(s_obs if s_obs is not None else np.zeros(4))
        self.a = a + buffer
        self.buffer = buffer
        self.BUFFER_FLAG = True

    def __repr__(self):
        return (f"{type(self).__name__}(a = {self.a}, buffer = {self.buffer}, "
                f"buffer_applied: {self.BUFFER_FLAG} )\n s_obs = {self.s_obs}")

    def _grow(self, d):
        self.a += d

    @property
    def cx(self):
        return self.s_obs[0]

    @property
    def cy(self):
        return self.s_obs[1]

    def slot(self):
        return OBS_CONE, [self.s_obs[0], self.s_obs[1], self.s_obs[2], self.s_obs[3], self.a,
                          self.beta, 0.0, 0.0]

    def dbeta(self, **kwargs):
        """Equal to dtheta() as the reference defines it (cbf/obstacles.py:460-466)."""
        return self.dtheta()

    def update(self, s=None, s_obs=None, buffer=None, **kwargs):
        if "a" in kwargs:
            self.a = kwargs["a"]
        if s is not None:
            self.s = _state4(s)
        if s_obs is not None:
            self.s_obs = _state4(s_obs)
        if buffer is not None:
            self._set_buffer(buffer)
        if "beta" in kwargs:
            self.beta = kwargs["beta"]

    def get_half_angle(self):
        """Apex half angle of the cone (reference cbf/obstacles.py:507-510)."""
        d = float(np.hypot(self.s[0] - self.s_obs[0], self.s[1] - self.s_obs[1]))
        cb = ZERO_TOL
        if d > abs(self.a):
            cb = float(np.sqrt(d * d - self.a * self.a)) + ZERO_TOL
        return float(np.arccos(cb / d)) if d > ZERO_TOL else float(np.pi / 2)

    def update_by_bounding_box(self, bbox):
        if not isinstance(bbox, BoundingBox):
            raise TypeError("Expected an object of type cbf.obstacles.BoundingBox, but got "
                            + type(bbox).__name__)
        self.a = float(np.hypot(bbox.extent.x, bbox.extent.y)) + (self.buffer if self.BUFFER_FLAG else 0.0)
        self.update(s_obs=[bbox.location.x, bbox.location.y, 0.0, bbox.velocity])

    @classmethod
    def from_bounding_box(cls, s=None, bbox=None, buffer=0.5, **kwargs):
        if not isinstance(bbox, BoundingBox):
            raise TypeError("Expected an object of type cbf.obstacles.BoundingBox, but got "
                            + type(bbox).__name__)
        return cls(a=float(np.hypot(bbox.extent.x, bbox.extent.y)), s=s,
                   s_obs=[bbox.location.x, bbox.location.y, 0.0, bbox.velocity], buffer=buffer,
                   **kwargs)


class PolyLane(Obstacle2DBase):
    """Polynomial lane boundary y = c0 + c1 x + ... (up to degree 6; the reference takes a coefficient array of any
    length, cbf/obstacles.py:578-593); the barrier is the squared distance to the nearest curve point minus `buffer`
    (reference cbf/obstacles.py:545-773)."""
    type = Obstacle2DTypes.POLY_LANE

    def __init__(self, coefficients, s=None, s_obs=None, buffer=1.50, **kwargs):
        super().__init__()
        if "id" in kwargs:
            self.id = kwargs["id"]
        self.beta = kwargs.get("beta", 0.0)
        self.update_coeffs(coefficients)
        self.s = _state4(s if s is not None else np.zeros(4))
        self.s_obs = _state4(s_obs if s_obs is not None else np.zeros(4))
        self.buffer = buffer

    def update_coeffs(self, coefficients):
        """coefficients = [a0, a1, a2, ...] of f(x) = a0 + a1 x + a2 x^2 + ... (degree <= 6: an obstacle slot
        holds seven coefficients and the buffer)."""
        c = np.asarray(coefficients, dtype=np.float64).reshape(-1)
        if c.size > 7 and np.any(c[7:] != 0.0):
            raise ValueError("PolyLane supports polynomials up to degree 6, got degree %d" % (c.size - 1))
        self.coeffs = c
        self.order = int(c.size - 1)

    def evaluate_polynomial(self, x, **kwargs):
        return np.polynomial.polynomial.polyval(x, self.coeffs)

    def slot(self):
        c = np.zeros(7)
        n = min(7, self.coeffs.size)
        c[:n] = self.coeffs[:n]
        if np.any(c[4:] != 0.0):     # degree 4 ... 6: the kind with seven coefficients
            return OBS_LANE6, [c[0], c[1], c[2], c[3], c[4], c[5], c[6], self.buffer]
        return OBS_LANE, [c[0], c[1], c[2], c[3], self.buffer, 0.0, 0.0, 0.0]

    def update(self, s=None, s_obs=None, buffer=None, **kwargs):
        if s is not None:
            self.s = _state4(s)
        if s_obs is not None:
            self.s_obs = _state4(s_obs)
        if buffer is not None:
            self.buffer = buffer

    def update_coeffs_by_curve_fit(self, x_pts, y_pts, **kwargs):
        self.update_coeffs(self.fit_polynomial_curve(x_pts, y_pts, **kwargs).coeffs)

    @classmethod
    def fit_polynomial_curve(cls, x_pts, y_pts, n=3, x_fixed_pts=None, y_fixed_pts=None,
                             fixed_pts_idx=None, alpha=0.01, sigma=None, initial_coeffs=None):
        """Weighted least-squares polynomial through the points; points listed as fixed get
        the small standard deviation `alpha`, the others 10 (reference 715-773, which runs
        scipy.optimize.curve_fit on the same linear model)."""
        x = np.asarray(x_pts, dtype=np.float64).flatten()
        y = np.asarray(y_pts, dtype=np.float64).flatten()
        if x.size != y.size:
            raise ValueError("Incompatible array sizes for x points and y points. Received: %s and %s"
                             % (x.shape, y.shape))
        sig = np.full_like(x, 10.0) if sigma is None else np.asarray(sigma, dtype=np.float64).flatten()
        if (x_fixed_pts is None) != (y_fixed_pts is None):
            raise ValueError("Both fixed point arrays have to be specified.")
        if x_fixed_pts is not None:
            xf = np.asarray(x_fixed_pts, dtype=np.float64).flatten()
            yf = np.asarray(y_fixed_pts, dtype=np.float64).flatten()
            x, y = np.append(x, xf), np.append(y, yf)
            sig = np.append(sig, alpha * np.ones_like(xf))
        if fixed_pts_idx is not None:
            sig[fixed_pts_idx] = alpha
        V = np.vander(x, n + 1, increasing=True) / sig[:, None]
        coeffs, *_ = np.linalg.lstsq(V, y / sig, rcond=None)
        return cls(coeffs)


class ObstacleList2D(MutableMapping):
    """Ordered id -> obstacle mapping whose f/dx/dy/dtheta/dv/dt/dbeta stack the per-obstacle
    values in insertion order (reference cbf/obstacles.py:798-941)."""

    def __init__(self, data=()):
        self.mapping = {}
        self.update(data)
        self.timestamp = 0.0

    def __getitem__(self, key):
        return self.mapping[key]

    def __delitem__(self, key):
        del self.mapping[key]

    def __setitem__(self, key, value):
        if not isinstance(value, Obstacle2DBase):
            raise TypeError("Expected an object derived from Obstacle2DBase as value. Received "
                            + type(value).__name__)
        self.mapping[key] = value

    def __iter__(self):
        return iter(self.mapping)

    def __len__(self):
        return len(self.mapping)

    def __repr__(self):
        return f"{type(self).__name__}({self.mapping})"

    def set_timestamp(self, timestamp):
        self.timestamp = timestamp

    def update_by_bounding_box(self, bbox_dict=None, obs_type=Obstacle2DTypes.ELLIPSE2D, buffer=0.5):
        """Sync with a dict of bounding boxes: update known ids, create new ones, drop ids
        that left the scene (reference 833-858)."""
        if bbox_dict is None:
            return
        for key, bbox in bbox_dict.items():
            if key in self.mapping:
                self.mapping[key].update_by_bounding_box(bbox)
            elif obs_type == Obstacle2DTypes.ELLIPSE2D:
                self[key] = Ellipse2D.from_bounding_box(bbox=bbox, buffer=buffer, id=key)
            elif obs_type == Obstacle2DTypes.COLLISION_CONE2D:
                self[key] = CollisionCone2D.from_bounding_box(bbox=bbox, buffer=buffer, id=key)
        for key in list(self.mapping):
            if key not in bbox_dict:
                self.pop(key)

    def update_state(self, s, s_obs_dict=None, buffer=None, **kwargs):
        if s_obs_dict is None:
            for obstacle in self.mapping.values():
                obstacle.update(s=s, s_obs=None, buffer=buffer, **kwargs)
            return
        if not isinstance(s_obs_dict, dict):
            raise ValueError("Expected dictionary for obstacle dictionary")
        for obstacle in self.mapping.values():
            obstacle.update(s=s, buffer=buffer, **kwargs)
        for key, s_obs in s_obs_dict.items():
            if key in self.mapping:
                self.mapping[key].update(s_obs=s_obs)
            else:
                warnings.warn(f"Unknown key provided in s_obs_dict: {key}")

    # -- packing for the C ABI
    def slots(self):
        """kinds (M,) uint8 and slots (M, 8) float64 in insertion order."""
        kinds = np.empty(len(self.mapping), dtype=np.uint8)
        sl = np.zeros((len(self.mapping), 8))
        for i, obs in enumerate(self.mapping.values()):
            k, v = obs.slot()
            kinds[i] = k
            sl[i] = v
        return kinds, sl

    def _stack(self, col):
        if not self.mapping:
            return np.zeros((0, 1))
        first = next(iter(self.mapping.values()))
        kinds, sl = self.slots()
        return _device.partials(first.s, kinds, sl)[:, col].reshape(-1, 1)

    def f(self, *args, **kwargs):
        return self._stack(0)

    def dx(self, *args, **kwargs):
        return self._stack(1)

    def dy(self, *args, **kwargs):
        return self._stack(2)

    def dtheta(self, *args, **kwargs):
        return self._stack(3)

    def dv(self, *args, **kwargs):
        return self._stack(4)

    def dt(self, *args, **kwargs):
        return self._stack(5)

    def dbeta(self, *args, **kwargs):
        out = np.zeros((len(self.mapping), 1))
        for i, obs in enumerate(self.mapping.values()):
            out[i, 0] = obs.dbeta()
        return out

    def gradient(self, *args, **kwargs):
        """(M, 4) matrix [dx | dy | dtheta | dv]."""
        if not self.mapping:
            return np.zeros((0, 4))
        first = next(iter(self.mapping.values()))
        kinds, sl = self.slots()
        return _device.partials(first.s, kinds, sl)[:, 1:5]
