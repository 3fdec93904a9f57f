"""Constants and small helpers of the reference's cbf/utils.py that the filter API uses."""
import math

ZERO_TOL = 1e-3  # reference cbf/utils.py:27


def normalize_angle(angle):
    """Angle wrapped to [-pi, pi] (reference cbf/utils.py:93-106, which subtracts / adds 2 pi in a
    loop; here an angle of many turns is brought near first and a non-finite one returned as it
    is, where that loop would run for hours or for ever)."""
    if not math.isfinite(angle):
        return angle
    if abs(angle) > 64.0:
        angle = math.remainder(angle, 2.0 * math.pi)
    while angle > math.pi:
        angle -= 2.0 * math.pi
    while angle < -math.pi:
        angle += 2.0 * math.pi
    return angle


def saturation(x, x_min, x_max):
    """Clamp (reference cbf/utils.py:111-117)."""
    return x_max if x > x_max else x_min if x < x_min else x


def sigmoid(x):
    return 1.0 / (1.0 + math.exp(-x))


def get_closest_idx(x, x_list):
    """Index of the entry of x_list nearest to x (reference cbf/utils.py:119-121)."""
    best, bd = 0, float("inf")
    for i, xi in enumerate(x_list):
        d = abs(x - xi)
        if d < bd:
            best, bd = i, d
    return best


def vec_norm(x):
    """Euclidean norm of a vector given as any flat sequence (reference cbf/utils.py:123-124)."""
    return math.sqrt(sum(float(v) * float(v) for v in _flat(x)))


def _flat(x):
    try:
        it = iter(x)
    except TypeError:
        return [x]
    out = []
    for v in it:
        out.extend(_flat(v))
    return out


class TimerError(Exception):
    """Raised when a Timer timestamp decreases."""


class Timer:
    """Monotone timestamp holder (reference cbf/utils.py:33-48)."""

    def __init__(self, timestamp=0.0):
        self._timestamp = timestamp

    @property
    def timestamp(self):
        return self._timestamp

    @timestamp.setter
    def timestamp(self, value):
        if self._timestamp > value:
            raise TimerError("Negative time or decreasing timestamp detected.")
        self._timestamp = value
