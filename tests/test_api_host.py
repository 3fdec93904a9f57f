"""Host-side behaviour of the reference-compatible object API (no GPU needed)."""
import numpy as np
import pytest

from sccav_cbf_amd.cbf import install_as_cbf
from sccav_cbf_amd.cbf._vec import Point2, Vector2, Vector3
from sccav_cbf_amd.cbf.cbf import DBM_CBF_2DS, KBM_VC_CBF2D
from sccav_cbf_amd.cbf.controllers import PID1, LateralStanley
from sccav_cbf_amd.cbf.geometry import Rotation
from sccav_cbf_amd.cbf.obstacles import (BoundingBox, CollisionCone2D, Ellipse2D, Obstacle2DTypes,
                                         ObstacleList2D, PolyLane, RadialEllipse2D)
from sccav_cbf_amd.cbf.utils import ZERO_TOL, normalize_angle, vec_norm


def test_reference_import_paths():
    install_as_cbf()
    from cbf.cbf import DBM_CBF_2DS as D  # noqa: F401
    from cbf.controllers import LateralStanley as L  # noqa: F401
    from cbf.obstacles import CollisionCone2D as Cc, Ellipse2D as E, PolyLane as P  # noqa: F401
    from cbf.utils import ZERO_TOL as Z
    assert Z == 1e-3 == ZERO_TOL


def test_ellipse_constructor_and_buffer():
    e = Ellipse2D(a=20, b=10, center=Point2(3.0, 4.0), theta=0.2, buffer=0.5, id=7)
    assert (e.a, e.b, e.id, e.type) == (20.5, 10.5, 7, Obstacle2DTypes.ELLIPSE2D)
    with pytest.raises(TypeError):
        Ellipse2D(1, 1, center=(0, 0))
    e.remove_buffer()
    assert (e.a, e.b) == (20, 10)
    with pytest.warns(UserWarning):
        e.remove_buffer()
    e.apply_buffer()
    e.update(buffer=1.0)
    assert (e.a, e.b, e.buffer) == (21.0, 11.0, 1.0)
    e.update(b=3.0)
    assert e.b == 3.0 and e.a == 21.0
    e.update_velocity_by_magnitude(2.0)
    assert np.allclose([e.vel.x, e.vel.y], [2 * np.cos(0.2), 2 * np.sin(0.2)])
    e.update_orientation(0.0)
    assert np.allclose([e.vel.x, e.vel.y, e.theta], [2.0, 0.0, 0.0])
    kind, slot = e.slot()
    assert kind == 0 and slot[:7] == [3.0, 4.0, 21.0, 3.0, 0.0, 2.0, 0.0]
    # the ego state does not overwrite the obstacle's own heading / velocity
    e.update(s=[1, 2, 0.9, 5.0])
    assert e.theta == 0.0 and e.vel.x == 2.0


def test_cone_lane_radial_slots():
    c = CollisionCone2D(11.0, [0, 5, 0.3, 10], [75, -35, 0.1, 2.0])
    assert c.a == 12.5 and c.slot() == (1, [75.0, -35.0, 0.1, 2.0, 12.5, 0.0, 0.0, 0.0])
    c.update(s_obs=[1, 2, 3, 4], beta=0.05, buffer=0.5)
    assert c.a == 11.5 and c.slot()[1][:6] == [1.0, 2.0, 3.0, 4.0, 11.5, 0.05]
    assert 0 < c.get_half_angle() <= np.pi / 2
    lane = PolyLane(np.array([19.4, 0.0]), s=[0, 15, 0, 3])
    assert lane.order == 1 and lane.slot() == (2, [19.4, 0.0, 0.0, 0.0, 1.5, 0.0, 0.0, 0.0])
    assert np.allclose(lane.evaluate_polynomial(np.array([0.0, 2.0])), [19.4, 19.4])
    quintic = PolyLane(np.array([1.0, 0.2, -0.03, 0.004, 5e-4, -6e-5]), buffer=0.8)       # degree 4 ... 6: the long kind
    assert quintic.order == 5 and quintic.slot() == (8, [1.0, 0.2, -0.03, 0.004, 5e-4, -6e-5, 0.0, 0.8])
    assert PolyLane(np.array([3.0, 0.1, 0, 0, 0, 0, 0])).slot()[0] == 2                     # zeros beyond the cubic: kind 2
    with pytest.raises(ValueError):
        PolyLane(np.array([0, 0, 0, 0, 0, 0, 0, 1.0]))
    r = RadialEllipse2D(1.7, 1.7, center=Vector2(5, 5), kv=1.0)
    r.update_velocity(Vector2(-1, -1))
    assert r.slot() == (3, [5, 5, 1.7, 1.7, -1, -1, 1.0, 0.0])


def test_lane_fit_known_answers():
    """A cubic through the four points of the reference's lane_cbf_test.py:199-200
    interpolates them exactly; straight lanes of the CARLA script come out as lines."""
    x = np.array([[-1.371, -0.75, 0, 0.333]])
    y = np.array([[0, 6.938, 3, 1.852]])
    lane = PolyLane.fit_polynomial_curve(x, y, n=3)
    assert np.allclose(lane.evaluate_polynomial(x.ravel()), y.ravel(), atol=1e-9)
    xs = np.linspace(-95.0, 0, 100)
    line = PolyLane.fit_polynomial_curve(xs, np.full_like(xs, 19.4), n=1)
    assert np.allclose(line.coeffs, [19.4, 0.0], atol=1e-9)
    with pytest.raises(ValueError):
        PolyLane.fit_polynomial_curve([0, 1], [0, 1, 2])
    fixed = PolyLane.fit_polynomial_curve([0, 1, 2, 3], [0, 1.2, 1.9, 3.1], n=1, x_fixed_pts=[10.0], y_fixed_pts=[0.0])
    assert abs(fixed.evaluate_polynomial(10.0)) < 1e-3


def test_obstacle_list_semantics():
    ol = ObstacleList2D()
    with pytest.raises(TypeError):
        ol["x"] = 3
    ol.update({1: Ellipse2D(2, 1, Vector2(0, 0)), "lane": PolyLane(np.array([1.0, 0.0]))})
    ol[5] = CollisionCone2D(1.0, [0, 0, 0, 1], [5, 5, 0, 0])
    assert len(ol) == 3 and list(ol) == [1, "lane", 5]
    kinds, slots = ol.slots()
    assert list(kinds) == [0, 2, 1] and slots.shape == (3, 8)
    ol.update_state(s=[1, 2, 0.5, 3.0])
    assert all(np.allclose(o.s, [1, 2, 0.5, 3.0]) for o in ol.values())
    ol.update_state(s=[1, 2, 0.5, 3.0], s_obs_dict={5: [9, 9, 0, 0]})
    assert ol[5].cx == 9
    with pytest.raises(ValueError):
        ol.update_state(s=[1, 2, 0.5, 3.0], s_obs_dict=[1, 2])
    ol.pop("lane")
    assert len(ol) == 2


def test_bounding_box_ingestion():
    bb = BoundingBox(extent=Vector3(2.0, 1.0, 0.8), location=Vector3(10, -3, 0), rotation=Rotation(yaw=0.4), velocity=3.0)
    ol = ObstacleList2D()
    ol.update_by_bounding_box({42: bb}, obs_type=Obstacle2DTypes.ELLIPSE2D, buffer=0.5)
    e = ol[42]
    assert (e.a, e.b, e.theta, e.center.x, e.center.y) == (2.5, 1.5, 0.4, 10, -3)
    bb2 = BoundingBox(extent=Vector3(3.0, 1.0, 0.8), location=Vector3(11, -3, 0), rotation=Rotation(yaw=0.5))
    ol.update_by_bounding_box({42: bb2, 43: bb})
    assert ol[42].a == 3.5 and ol[42].center.x == 11 and 43 in ol
    ol.update_by_bounding_box({43: bb})
    assert list(ol) == [43]
    oc = ObstacleList2D()
    oc.update_by_bounding_box({1: bb}, obs_type=Obstacle2DTypes.COLLISION_CONE2D, buffer=0.5)
    assert abs(oc[1].a - (np.hypot(2, 1) + 0.5)) < 1e-12 and oc[1].s_obs[3] == 3.0
    with pytest.raises(TypeError):
        Ellipse2D.from_bounding_box(bbox="no")


def test_filter_object_errors_and_setters():
    f = DBM_CBF_2DS(alpha=2.0)
    f.set_model_params(lr=1.2, lf=1.6)
    f.set_alpha(0.5)
    with pytest.raises(ValueError):
        f.set_qp_cost_weight(np.eye(3))
    f.set_qp_cost_weight(np.diag([0.5, 0.5]))
    f.update_state(s=[0, 0, 0.1, 4.0])
    with pytest.raises(ValueError):
        f.solve_cbf([0.0, 0.0])                      # empty obstacle list (cbf/cbf.py:177-180)
    assert f.gc().shape == (4, 2) and f.fc().shape == (4, 1)
    assert np.allclose(f.gc()[:, 1], [-4 * np.sin(0.1), 4 * np.cos(0.1), 4 / 1.2, 0])
    p = f._params()
    assert (p.alpha, p.lr, p.lf, p.dynamics) == (0.5, 1.2, 1.6, 0)
    k = KBM_VC_CBF2D()
    k.set_model_params(L=2.9)
    k.update_state(Point2(1, 2), theta=0.3)
    with pytest.raises(ValueError):
        k.solve_cbf([1.0, 0.0])


def test_controllers():
    st = LateralStanley(lr=1.45, lf=1.45, k=0.5, ks=0.0)
    xs = np.linspace(0, 50, 501)
    traj = [(x, 0.0, 0.0, 8.0) for x in xs]
    st.set_trajectory(traj)
    st.update_state(10.0, 1.0, 0.0, 5.0)
    delta, idx = st.control()
    # ego 1 m to the left of a straight course: steer right by atan(k e / v)
    assert idx == int(round((10.0 + 1.45) / 0.1))
    assert abs(delta + np.arctan2(0.5 * 1.0, 5.0)) < 1e-9
    st.update_state(5.0, 1.0, 0.0, 5.0)
    _, idx2 = st.control()
    assert idx2 == idx                               # the target index never moves backwards
    with pytest.raises(TypeError):
        st.control(front_coords=(1, 2))
    pid = PID1(kp=2.0, kd=0.5, ki=0.1)
    pid.set_dt(0.1)
    u1 = pid.control(1.0, 3.0)
    assert abs(u1 - (2.0 * 2 + 0.1 * 0.2 + 0.5 * 20)) < 1e-12
    u2 = pid.control(2.0, 3.0)
    assert abs(u2 - (2.0 * 1 + 0.1 * 0.3 + 0.5 * (-10))) < 1e-12
    assert abs(normalize_angle(3 * np.pi) - np.pi) < 1e-12 and abs(vec_norm([3, 4]) - 5) < 1e-12
    assert abs(normalize_angle(1e17)) <= np.pi and normalize_angle(float("inf")) == float("inf")   # and they return


def _closed_loop_parts(A=8, M=3, N=5, dtype=None):
    import torch
    dtype = dtype or torch.float32
    return dict(traj=torch.zeros((N, 4), dtype=dtype), state=torch.zeros((A, 4), dtype=dtype),
                obs=torch.zeros((M, 2, A, 4), dtype=dtype), obs_type=torch.zeros((M,), dtype=torch.uint8))


def test_closed_loop_rejects_mismatched_tensors_before_any_launch():
    """Tensors built for another batch size, precision or layout must raise ValueError on the host:
    the C ABI checks pointers only, the kernels index by the shapes they are told."""
    import torch
    from sccav_cbf_amd import _lib
    from sccav_cbf_amd.batched import BatchedCBF
    from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
    flt, nom = BatchedCBF(), BatchedNominal()
    ok = _closed_loop_parts()
    loop = ClosedLoop(flt, nom, **ok)                  # consistent CPU tensors: accepted by the checks ...
    with pytest.raises(_lib.SccbfError):               # ... and refused only for want of a GPU
        loop.rollout(3)
    bad = [("obs", torch.zeros((3, 2, 8, 4), dtype=torch.float64)),           # precision of another batch
           ("obs", torch.zeros((3, 2, 9, 4))),                               # another A
           ("obs", torch.zeros((1, 3, 2, 32, 4))),                           # tiles of the wrong size
           ("obs", torch.zeros((3, 2, 4, 8)).transpose(2, 3)),               # a view, not contiguous
           ("obs_type", torch.zeros((3, 7), dtype=torch.uint8)),             # per-agent kinds of another A
           ("obs_type", torch.zeros((3,), dtype=torch.int32)),
           ("traj", torch.zeros((5, 3))),
           ("traj", torch.zeros((5, 4), dtype=torch.float64)),
           ("state", torch.zeros((8, 3)))]
    for name, t in bad:
        with pytest.raises(ValueError):
            ClosedLoop(flt, nom, **dict(ok, **{name: t}))
    # tensors swapped in after construction are checked again by rollout() and tick()
    loop.obs = torch.zeros((3, 2, 9, 4))
    with pytest.raises(ValueError):
        loop.rollout(3)
    with pytest.raises(ValueError):
        loop.tick()
    loop.obs = ok["obs"]
    loop.target_idx = torch.zeros((7,), dtype=torch.int32)
    with pytest.raises(ValueError):
        loop.rollout(3)
    with pytest.raises(ValueError):
        ClosedLoop(flt, None, **ok)                    # a course without a controller


def test_closed_loop_refuses_slip_coordinates():
    """tick() and rollout() step the vehicle with steering angles; a filter set to slip-angle I/O would
    hand them something else."""
    from sccav_cbf_amd import _lib
    from sccav_cbf_amd.batched import BatchedCBF
    from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
    loop = ClosedLoop(BatchedCBF(dynamics="dbm", slip_io=True), BatchedNominal(), **_closed_loop_parts())
    with pytest.raises(ValueError, match="slip_io"):
        loop.tick()
    with pytest.raises(ValueError, match="slip_io"):
        loop.rollout(2)
    for dyn in ("kbm", "dum", "sadbm"):             # every model has its step: only the GPU is missing here
        loop = ClosedLoop(BatchedCBF(dynamics=dyn), BatchedNominal(), **_closed_loop_parts())
        with pytest.raises(_lib.SccbfError):
            loop.tick()


def test_rollout_leaves_the_filter_flags_alone():
    from sccav_cbf_amd import _lib
    from sccav_cbf_amd.batched import BatchedCBF
    from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
    flt = BatchedCBF()
    flt.params.flags = _lib.FLAG_TYPE_PER_AGENT
    loop = ClosedLoop(flt, BatchedNominal(), **_closed_loop_parts())
    with pytest.raises(_lib.SccbfError):
        loop.rollout(2)
    assert flt.params.flags == _lib.FLAG_TYPE_PER_AGENT


def test_library_version_matches_the_binding():
    from sccav_cbf_amd import _lib
    import re, os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sccbf.h")).read()
    assert int(re.search(r"#define SCCBF_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION
    assert _lib.load().sccbf_version() == _lib.ABI_VERSION
