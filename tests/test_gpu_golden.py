"""The HIP path against the frozen oracle vectors of tests/golden/oracle_cases.npz (through the C ABI):
barrier values and partials of every obstacle kind including the degenerate cases, the filter of every vehicle
model, hand-made and random QPs.  Needs an MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sccav_cbf_amd import _lib  # noqa: E402
from sccav_cbf_amd.batched import BatchedCBF  # noqa: E402

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cases.npz"))
R_HALF = np.diag([0.5, 0.5])


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)], ids=["f64", "f32"])
def test_barriers_and_partials_of_every_kind(dtype, tol):
    """One agent per case, one slot each, kinds per agent: sccbf_partials_* gives h and its five partials."""
    kind, state, slot, want = G["bar_kind"], G["bar_state"], G["bar_slot"], G["bar_out"]
    K = len(kind)
    obs = np.ascontiguousarray(slot.reshape(1, K, 2, 4).transpose(0, 2, 1, 3))      # (M=1, 2, A=K, 4)
    out = torch.empty((1, 2, K, 4), dtype=dtype, device=DEV)
    lib = _lib.load()
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    p.flags = _lib.FLAG_TYPE_PER_AGENT
    ts, to, tk = _t(state, dtype), _t(obs, dtype), torch.from_numpy(kind.reshape(1, K).copy()).to(DEV)
    fn = lib.sccbf_partials_f64 if dtype == torch.float64 else lib.sccbf_partials_f32
    _lib.check(fn(C.byref(p), K, 1, ts.data_ptr(), to.data_ptr(), tk.data_ptr(), out.data_ptr(), None))
    torch.cuda.synchronize()
    got = out.double().cpu().numpy()[0].transpose(1, 0, 2).reshape(K, 8)[:, :6]
    if dtype == torch.float32:
        # the oracle on the inputs as fp32 holds them; a cone at the ego's own velocity divides by the
        # relative speed + ZERO_TOL and a lane's 2 / eta by eta: compared to the size of the case's largest partial
        from oracle import oracle as orc
        s32, o32 = state.astype(np.float32).astype(np.float64), slot.astype(np.float32).astype(np.float64)
        want = np.stack([[orc.barrier(int(k), s, o)[n] for n in ("h", "hx", "hy", "hth", "hv", "ht")]
                         for k, s, o in zip(kind, s32, o32)])
    scale = 1.0 + np.abs(want).max(axis=1, keepdims=True)
    err = np.abs(got - want) / scale
    if dtype == torch.float64:
        assert err.max() <= tol, (int(err.max(axis=1).argmax()), err.max())
    else:
        # three cases sit on a discontinuity of the formulas (d = a, d = 0, relative speed 0 through beta), where
        # single precision may fall on the other side; every other case within the tolerance
        bad = np.where(err.max(axis=1) > tol)[0]
        assert len(bad) <= 3, (bad, err.max(axis=1)[bad])


def _filter(model):
    kw = dict(alpha=1.0, R=R_HALF)
    if model == "dbm_slip_io":
        return BatchedCBF(slip_io=True, **kw)
    return BatchedCBF(dynamics=model, **kw)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("model", ["dbm", "dbm_slip_io", "kbm", "dum", "sadbm"])
@pytest.mark.parametrize("name", ["mixed16", "lanes2_cones6", "lanes6_cones6"])
def test_filter_of_every_model(name, model, dtype):
    state, obs, ty = G[name + "_state"], G[name + "_obs"], G[name + "_obs_type"]
    u_ref = G[name + "_u_ref_dum"] if model == "dum" else G[name + "_u_ref"]
    uo, so, ro = G["%s_%s_u" % (name, model)], G["%s_%s_status" % (name, model)], G["%s_%s_rows" % (name, model)]
    f = _filter(model)
    args = (_t(state, dtype), _t(u_ref, dtype), _t(obs, dtype), torch.from_numpy(ty).to(DEV))
    if model == "sadbm":
        slip = _t(G[name + "_slip_in"], dtype)
        ug, sg, rg = f.solve(*args, want_rows=True, slip=slip, dt=0.05, cone_beta_state=True)
    else:
        ug, sg, rg = f.solve(*args, want_rows=True)
    torch.cuda.synchronize()
    ug, sg, rg = ug.double().cpu().numpy(), sg.cpu().numpy(), rg.double().cpu().numpy()
    rerr = np.abs(rg - ro) / (np.abs(ro).max(axis=2, keepdims=True) + 1.0)
    if dtype == torch.float64:
        assert (sg == so).all()
        assert rerr.max() <= 1e-9
        assert np.abs(ug - uo).max() <= 1e-8 * (1 + np.abs(uo).max())
        if model == "sadbm":
            assert np.abs(slip.cpu().numpy() - G[name + "_sadbm_slip_out"]).max() <= 1e-9
    else:
        assert (sg != so).sum() <= 1
        assert rerr.max() <= 1e-3
        both = (sg == 0) & (so == 0) & (np.abs(uo[:, 1]) < 1.2)
        scale = np.maximum(1.0, np.hypot(uo[:, 0], uo[:, 1]))
        assert np.quantile((np.abs(ug - uo).max(axis=1) / scale)[both], 0.95) <= 2e-3


@pytest.mark.parametrize("lanes", (0, 1, 4, 16, 64))
def test_qp_stage_on_frozen_rows(lanes):
    for tag, R in (("hand", np.eye(2)), ("rand", np.array([[2.0, 0.3], [0.3, 1.0]]))):
        rows, un = G["qp_%s_rows" % tag], G["qp_%s_u_nom" % tag]
        uo, so = G["qp_%s_u" % tag], G["qp_%s_status" % tag]
        f = BatchedCBF(alpha=1.0, R=R, lanes_per_qp=lanes)
        ug, sg = f.solve_qp(_t(rows, torch.float64), _t(un, torch.float64))
        torch.cuda.synchronize()
        assert (sg.cpu().numpy() == so).all(), tag
        assert np.abs(ug.cpu().numpy() - uo).max() <= 1e-6 * (1 + np.abs(uo).max()), tag
