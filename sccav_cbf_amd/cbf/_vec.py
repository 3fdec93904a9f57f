"""Minimal 2-D / 3-D vector types for the obstacle API.

The reference takes euclid.Vector2 / Point2 / Vector3 objects (cbf/obstacles.py:25,
146-156).  When the `euclid` package is importable its classes are used as they are;
otherwise these small stand-ins provide the attributes the obstacle classes read
(x, y, z, magnitude(), copy()).
"""
import math

try:  # pragma: no cover - euclid is not installed in the build image
    from euclid import Point2, Vector2, Vector3  # noqa: F401
    HAVE_EUCLID = True
except Exception:  # ModuleNotFoundError, or a euclid written for another Python
    HAVE_EUCLID = False

    class Vector2:
        __slots__ = ("x", "y")

        def __init__(self, x=0.0, y=0.0):
            self.x = x
            self.y = y

        def copy(self):
            return self.__class__(self.x, self.y)

        def magnitude(self):
            return math.hypot(self.x, self.y)

        def __abs__(self):
            return self.magnitude()

        def __iter__(self):
            return iter((self.x, self.y))

        def __eq__(self, other):
            return hasattr(other, "x") and hasattr(other, "y") and self.x == other.x and self.y == other.y

        def __add__(self, o):
            return self.__class__(self.x + o.x, self.y + o.y)

        def __sub__(self, o):
            return self.__class__(self.x - o.x, self.y - o.y)

        def __mul__(self, k):
            return self.__class__(self.x * k, self.y * k)

        __rmul__ = __mul__

        def __truediv__(self, k):
            return self.__class__(self.x / k, self.y / k)

        def __repr__(self):
            return "%s(%.2f, %.2f)" % (type(self).__name__, self.x, self.y)

    class Point2(Vector2):
        __slots__ = ()

    class Vector3:
        __slots__ = ("x", "y", "z")

        def __init__(self, x=0.0, y=0.0, z=0.0):
            self.x = x
            self.y = y
            self.z = z

        def copy(self):
            return self.__class__(self.x, self.y, self.z)

        def magnitude(self):
            return math.sqrt(self.x * self.x + self.y * self.y + self.z * self.z)

        def __eq__(self, other):
            return (hasattr(other, "z") and self.x == other.x and self.y == other.y
                    and self.z == other.z)

        def __repr__(self):
            return "Vector3(%.2f, %.2f, %.2f)" % (self.x, self.y, self.z)


def is_vec2(v):
    return hasattr(v, "x") and hasattr(v, "y") and not isinstance(v, (tuple, list))
