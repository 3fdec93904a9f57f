"""Reference-compatible object API: cbf.cbf, cbf.obstacles, cbf.controllers, cbf.utils."""
from . import utils  # noqa: F401


def install_as_cbf():
    """Register this package as the top-level module `cbf`, so that code written against the
    reference (`from cbf.obstacles import Ellipse2D`, `from cbf.cbf import DBM_CBF_2DS`)
    imports these classes."""
    import importlib
    import sys
    pkg = sys.modules[__name__]
    sys.modules["cbf"] = pkg
    for sub in ("utils", "geometry", "obstacles", "cbf", "controllers"):
        sys.modules["cbf." + sub] = importlib.import_module(__name__ + "." + sub)
    return pkg
