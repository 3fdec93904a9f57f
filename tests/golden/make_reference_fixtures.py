#!/usr/bin/env python3
"""Generate the golden fixtures that come from the reference tree itself.

Run once in the build container (where /root/reference is mounted); the
outputs are small data files committed next to this script.  Nothing here is
needed at test time on the GPU box.

  sce_course.npz      cubic-spline course of
                      test_scripts/stanley_controller_ellipse.py:584-588,
                      produced by importing the reference's own
                      PathPlanning/CubicSpline/cubic_spline_planner.py
                      (numpy/math/bisect only) and calling
                      calc_spline_course(ax, ay, ds=0.1).
  beta_vs_time.npz    the 277 (t, beta_deg) samples stored in
                      test_scripts/beta_vs_time.mat, written by
                      stanley_controller_ellipse.py:1066-1069.
"""
import hashlib
import os
import sys

import numpy as np
from scipy.io import loadmat

REF = "/root/reference/test_scripts"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, os.path.join(REF, "PathPlanning", "CubicSpline"))
    import cubic_spline_planner as csp

    ax = [0.0, 100.0, 100.0, 50.0, 60.0]      # stanley_controller_ellipse.py:584
    ay = [0.0, 0.0, -30.0, -20.0, 0.0]        # stanley_controller_ellipse.py:585
    cx, cy, cyaw, ck, s = csp.calc_spline_course(ax, ay, ds=0.1)
    arr = np.array([cx, cy, cyaw])
    print("course points:", arr.shape[1], "sha256:", hashlib.sha256(arr.tobytes()).hexdigest()[:16])
    np.savez_compressed(os.path.join(HERE, "sce_course.npz"),
                        cx=np.asarray(cx), cy=np.asarray(cy), cyaw=np.asarray(cyaw),
                        knots_x=np.asarray(ax), knots_y=np.asarray(ay))

    m = loadmat(os.path.join(REF, "beta_vs_time.mat"))
    np.savez_compressed(os.path.join(HERE, "beta_vs_time.npz"),
                        t_arr=m["t_arr"].ravel(), beta_deg=m["beta_deg"].ravel())
    print("beta samples:", m["beta_deg"].size)


if __name__ == "__main__":
    main()
