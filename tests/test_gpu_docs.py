"""The usage examples of the integration guide, run as a user would: in a fresh process, written the way the guide
writes them (names filled in), every result checked for shape and sanity.  Needs an MI355X."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np

# ---- 1. the reference's call pattern (no torch in the script)
import sccav_cbf_amd.cbf as amd_cbf
amd_cbf.install_as_cbf()
from cbf.obstacles import CollisionCone2D, Ellipse2D, PolyLane
from cbf.cbf import DBM_CBF_2DS
from cbf.controllers import LateralStanley, PID1
lr = lf = 1.45
s = np.array([0.0, 5.0, 0.35, 10.0])
s_obs = np.array([30.0, 12.0, 0.0, 0.0])
cbf_controller = DBM_CBF_2DS(alpha=1)
cbf_controller.set_model_params(lr=lr, lf=lf)
cbf_controller.obstacle_list2d.update({0: CollisionCone2D(11.18, s, s_obs)})
cbf_controller.update_state(s=s)
cbf_controller.set_qp_cost_weight(np.diag([0.5, 0.5]))
u = cbf_controller.solve_cbf(np.array([0.3, 0.1]))
assert len(u) == 2 and np.isfinite([u[0], u[1]]).all() and cbf_controller.last_status == 0
info, u2 = cbf_controller.solve_cbf(np.array([0.3, 0.1]), return_solver=True)
assert np.allclose(info["x"], u2) and info["rows"].shape == (1, 4)

# ---- 2. batched API
import torch
from sccav_cbf_amd import BatchedCBF
from sccav_cbf_amd.batched import cone_slots, ellipse_slots, pack_obs, pack_obs_tiled, planes_to_tiled, capture_independent
dev = torch.device("cuda:0")
A = 1000
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda lo, hi, *shape: (lo + (hi - lo) * torch.rand(*shape, generator=g)).to(dev)
state = torch.stack([r(-50, 50, A), r(-50, 50, A), r(-3, 3, A), r(2, 12, A)], dim=1).contiguous()
u_ref = torch.stack([r(-1, 1, A), r(-0.3, 0.3, A)], dim=1).contiguous()
cx, cy = state[None, :, 0] + r(20, 60, 3, A), state[None, :, 1] + r(-30, 30, 3, A)
ox, oy = state[None, :, 0] + r(-60, -20, 2, A), state[None, :, 1] + r(-30, 30, 2, A)
slots = torch.cat([ellipse_slots(cx, cy, r(2, 20, 3, A), r(1, 10, 3, A), r(-3, 3, 3, A), r(-2, 2, 3, A), r(-2, 2, 3, A)),
                   cone_slots(ox, oy, r(-3, 3, 2, A), r(0, 5, 2, A), r(2, 6, 2, A))])
assert slots.shape == (5, A, 8)
obs = pack_obs(slots)
obs_type = torch.tensor([0, 0, 0, 1, 1], dtype=torch.uint8, device=dev)
flt = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]))
u, status = flt.solve(state, u_ref, obs, obs_type)
torch.cuda.synchronize()
assert u.shape == (A, 2) and status.shape == (A,) and int((status == 2).sum()) == 0
ut, st = flt.solve(state, u_ref, pack_obs_tiled(slots), obs_type)
assert torch.equal(ut, u) and torch.equal(st, status) and torch.equal(planes_to_tiled(obs), pack_obs_tiled(slots))
slip = torch.zeros((A, 2), device=dev)
us, ss = BatchedCBF(dynamics="sadbm", R=np.diag([0.5, 0.5])).solve(state, u_ref, obs, obs_type, slip=slip, dt=0.1, cone_beta_state=True)
assert us.shape == (A, 2)
u0, s0 = torch.empty_like(u), torch.empty_like(status)
u1, s1 = torch.empty_like(u), torch.empty_like(status)
flt.solve(state, u_ref, obs, obs_type, u_out=u0, status=s0)
gr = capture_independent([lambda: flt.solve(state, u_ref, obs, obs_type, u_out=u0, status=s0),
                          lambda: flt.solve(state, u_ref, obs, obs_type, u_out=u1, status=s1)], branches=3)
gr.replay()
torch.cuda.synchronize()
assert torch.equal(u0, u) and torch.equal(u1, u)
po, pt = flt.prepare_obs(obs.clone(), obs_type.clone())
up, sp = flt.solve(state, u_ref, po, pt)
assert (sp != status).sum() <= 1 and float((up - u).abs().max()) < 1e-2 * (1 + float(u.abs().max()))

# ---- several batches in one launch, per-agent lists, least violation
from sccav_cbf_amd import batched
from sccav_cbf_amd.batched import BatchTable, planes_to_agents
batched.prewarm()
table = BatchTable([dict(state=state, u_ref=u_ref, obs=obs, obs_type=obs_type),
                    dict(state=state[:300].contiguous(), u_ref=u_ref[:300].contiguous(), obs=obs[:, :, :300].contiguous(), obs_type=obs_type)])
flt.solve_batches(table)
torch.cuda.synchronize()
assert torch.equal(table.batches[0]["u_out"], u) and torch.equal(table.batches[1]["u_out"], u[:300])
for b in table.batches:
    b["u_out"].zero_()
flt.solve_batches_static(table)
torch.cuda.synchronize()
assert torch.equal(table.batches[0]["u_out"], u) and torch.equal(table.batches[1]["status"], status[:300])
ul, sl = flt.solve(state, u_ref, planes_to_agents(obs), obs_type)      # obs as (A, M, 2, 4)
assert torch.equal(ul, u) and torch.equal(sl, status)
umv, smv, rows_mv = BatchedCBF(alpha=1.0, lr=1.45, lf=1.45, R=np.diag([0.5, 0.5]), min_violation=True).solve(state, u_ref, obs, obs_type)
torch.cuda.synchronize()
assert torch.equal(smv, status) and torch.equal(umv[status == 0], u[status == 0]) and rows_mv.shape == (5, A, 4)
cbf_controller.infeasible_policy = "min_violation"
assert len(cbf_controller.solve_cbf(np.array([0.3, 0.1]))) == 2
lane7 = PolyLane([4.0, 0.01, 1e-3, 1e-5, 1e-7, 0.0, 1e-11], s, buffer=1.5)       # degree 6
assert np.isfinite(lane7.f())

# ---- closed loop
from sccav_cbf_amd.nominal import BatchedNominal, ClosedLoop
T = 12
xs = torch.linspace(0, 200, 400, device=dev)
traj = torch.stack([xs, 5 * torch.sin(xs / 30), torch.atan(torch.cos(xs / 30) / 6), torch.full_like(xs, 30 / 3.6)], dim=1).contiguous()
nom = BatchedNominal(axle=2.9, k=0.5, ks=0.0, kp=1.0, target_speed=30 / 3.6)
loop = ClosedLoop(flt, nom, traj, state.clone(), obs, obs_type, max_steer=np.radians(30), dt=0.1)
for _ in range(T):
    loop.tick()
torch.cuda.synchronize()
assert loop.state.shape == (A, 4) and loop.u.shape == (A, 2) and loop.target_idx.shape == (A,)
loop2 = ClosedLoop(flt, nom, traj, state.clone(), obs, obs_type, max_steer=np.radians(30), dt=0.1)
stats = loop2.rollout(T, motion="constant_velocity", trace=True)
assert stats["trace"].shape == (T, A, 4) and stats["h_min"].shape == (A,)
loop2.rollout(T, motion="constant_velocity", out=stats)
loop3 = ClosedLoop(flt, None, None, state.clone(), obs, obs_type, dt=1 / 30)
loop3.rollout(5, motion="seeker")
torch.cuda.synchronize()

# ---- perception to slots
from sccav_cbf_amd.ingest import bbox_as_obs, ingest_bbox, lane_fit
M = 4
bbox = torch.zeros((M, 2, A, 4), device=dev)
bbox[:, 0, :, 0], bbox[:, 0, :, 1] = state[None, :, 0] + r(15, 50, M, A), state[None, :, 1] + r(-20, 20, M, A)
bbox[:, 0, :, 2], bbox[:, 0, :, 3] = r(1, 3, M, A), r(0.5, 1.5, M, A)
bbox[:, 1, :, 0], bbox[:, 1, :, 1] = r(-3, 3, M, A), r(0, 5, M, A)
kind = torch.tensor([0, 1, 0, 2], dtype=torch.uint8, device=dev)      # ellipse, cone, ellipse, and a lane slot to fit
ob = ingest_bbox(bbox, kind, buffer=0.5, tiled=True)
assert ob.shape == ((A + 63) // 64, M, 2, 64, 4)
K = 12
px = state[None, :, 0] + torch.linspace(-10, 40, K, device=dev)[:, None]
py = state[None, :, 1] + 4.0 + 0.01 * (px - state[None, :, 0]) ** 2
lane_fit(px.contiguous(), py.contiguous(), ob, 3, sigma=None, degree=3, buffer=1.5)
ui, si = flt.solve(state, u_ref, ob, kind)
bb, kb = bbox_as_obs(bbox.clone(), kind, buffer=0.5)
assert kb.tolist() == [6, 7, 6, 2]
torch.cuda.synchronize()
assert int((si == 2).sum()) == 0

# ---- 3. the ctypes binding as written
import ctypes as C
class SccbfParams(C.Structure):
    _fields_ = [("alpha", C.c_double), ("lr", C.c_double), ("lf", C.c_double), ("L", C.c_double),
                ("R", C.c_double * 4), ("zero_tol", C.c_double), ("dynamics", C.c_int32),
                ("lanes_per_qp", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_int32)]
lib = C.CDLL(%(lib)r)
lib.sccbf_solve_f64.restype = C.c_int
lib.sccbf_solve_f64.argtypes = [C.POINTER(SccbfParams), C.c_int64, C.c_int32] + [C.c_void_p] * 8
lib.sccbf_strerror.restype = C.c_char_p
p = SccbfParams(); lib.sccbf_default_params(C.byref(p)); p.alpha = 1.0; p.R[0] = p.R[3] = 0.5
state64, u_ref64, obs64 = state.double(), u_ref.double(), obs.double()
u_out, status64 = torch.empty((A, 2), dtype=torch.float64, device=dev), torch.empty((A,), dtype=torch.int32, device=dev)
rc = lib.sccbf_solve_f64(C.byref(p), A, 5, state64.data_ptr(), u_ref64.data_ptr(), obs64.data_ptr(),
                         obs_type.data_ptr(), u_out.data_ptr(), status64.data_ptr(), None,
                         torch.cuda.current_stream().cuda_stream)
assert rc == 0, lib.sccbf_strerror(rc)
torch.cuda.synchronize()
assert (status64 != status).sum() <= 1
print("DOCS OK")
'''


def test_integration_examples_run_as_written():
    from sccav_cbf_amd import _lib
    code = SCRIPT % {"root": ROOT, "lib": _lib.LIB}
    env = dict(os.environ, SCCBF_NO_REBUILD="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "DOCS OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
