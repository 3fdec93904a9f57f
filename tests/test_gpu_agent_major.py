"""Per-agent obstacle lists (SCCBF_FLAG_OBS_AGENT_MAJOR, obs (A, M, 2, 4)): the same bits as from planes, through
every kernel variant, every model, the fused-batch launch, the closed loop and the ingestion kernels.  Needs an
MI355X."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sccav_cbf_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu

R_HALF = np.diag([0.5, 0.5])
DEV = "cuda:0"


def _filter(lanes=0, dynamics="dbm", **kw):
    from sccav_cbf_amd.batched import BatchedCBF
    return BatchedCBF(alpha=1.0, R=R_HALF, lanes_per_qp=lanes, dynamics=dynamics, **kw)


def _case(A, M, seed, dtype, per_agent=False):
    from tests.campaigns.fuzz_parity import random_case
    rng = np.random.default_rng(seed)
    state, u_ref, obs, kinds = random_case(rng, A, M)
    ty = np.ascontiguousarray(np.broadcast_to(kinds[:, None], (M, A))) if per_agent else kinds
    return dict(state=torch.from_numpy(state).to(DEV).to(dtype), u_ref=torch.from_numpy(u_ref).to(DEV).to(dtype),
                obs=torch.from_numpy(obs).to(DEV).to(dtype).contiguous(), obs_type=torch.from_numpy(ty).to(DEV))


def test_layout_is_read_from_the_shape():
    from sccav_cbf_amd.batched import obs_layout, planes_to_agents
    planes = torch.zeros((5, 2, 7, 4))
    assert obs_layout(planes, 7) == (0, 5, (5, 2, 7, 4))
    assert obs_layout(planes_to_agents(planes), 7) == (_lib.FLAG_OBS_AGENT_MAJOR, 5, (7, 5, 2, 4))
    two = torch.zeros((7, 2, 2, 4))                       # seven agents with two slots each, or seven slots of two agents
    assert obs_layout(two, 7)[0] == _lib.FLAG_OBS_AGENT_MAJOR and obs_layout(two, 2)[0] == 0
    assert obs_layout(two)[0] == 0


@pytest.mark.parametrize("lanes", (0, 1, 4, 8, 16, 32, 64))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_agent_major_equals_planes(lanes, dtype):
    """u_out, status and rows_out bit for bit, for row counts on both sides of every kernel family's limit and
    agent counts that leave workgroups partly filled."""
    from sccav_cbf_amd.batched import planes_to_agents
    for M in (1, 16, 37, 129, 257, 1024):
        if lanes not in (0, 16, 32, 64) and M > 16 * lanes:
            continue
        for A, per_agent in ((1, False), (67, True), (1000, False)):
            if M == 1024 and A == 1000 and lanes not in (0, 64):
                continue
            flt = _filter(lanes=lanes, slip_io=True)
            t = _case(A, M, 3 * M + A, dtype, per_agent)
            try:
                want = flt.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], want_rows=True)
            except _lib.SccbfError as exc:      # this many rows do not fit the LDS of so narrow a variant
                assert "lanes_per_qp" in str(exc) and M > 128
                continue
            got = flt.solve(t["state"], t["u_ref"], planes_to_agents(t["obs"]), t["obs_type"], want_rows=True)
            torch.cuda.synchronize()
            for w, g, name in zip(want, got, ("u_out", "status", "rows_out")):
                assert torch.equal(w, g), (name, M, A, lanes)


@pytest.mark.parametrize("dynamics", ["kbm", "dum", "sadbm"])
def test_agent_major_in_the_other_models(dynamics):
    from sccav_cbf_amd.batched import planes_to_agents
    for M, A in ((16, 500), (300, 130)):
        flt = _filter(dynamics=dynamics)
        t = _case(A, M, 11 + M, torch.float32)
        kw = {}
        if dynamics == "sadbm":
            kw = dict(dt=0.05)
        outs = []
        for obs in (t["obs"], planes_to_agents(t["obs"])):
            if dynamics == "sadbm":
                kw["slip"] = torch.full((A, 2), 0.01, device=DEV)
            outs.append(flt.solve(t["state"], t["u_ref"], obs, t["obs_type"], want_rows=True, **kw))
        torch.cuda.synchronize()
        for w, g in zip(*outs):
            assert torch.equal(w, g), (dynamics, M)


def test_agent_major_in_fused_batches():
    from sccav_cbf_amd.batched import BatchTable, planes_to_agents
    for M, lanes in ((16, 0), (300, 0), (1024, 0)):
        flt = _filter(lanes=lanes, slip_io=True)
        cases = [_case(A, M, 50 + A, torch.float32) for A in (9, 700, 64)]
        want = [flt.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"]) for t in cases]
        for t in cases:
            t["obs"] = planes_to_agents(t["obs"])
        table = BatchTable(cases)
        assert table.flags & _lib.FLAG_OBS_AGENT_MAJOR
        flt.solve_batches(table)
        torch.cuda.synchronize()
        for b, w in zip(table.batches, want):
            assert torch.equal(b["u_out"], w[0]) and torch.equal(b["status"], w[1])


def test_ingestion_writes_agent_major():
    """Boxes -> slots, as-written -> prepared, lane points -> lane slot: each kernel writes the per-agent lists that
    its planes output permutes to."""
    from sccav_cbf_amd.batched import planes_to_agents, prepare_obs
    from sccav_cbf_amd.ingest import ingest_bbox, lane_fit
    torch.manual_seed(5)
    A, M = 333, 12
    bbox = torch.rand((M, 2, A, 4), device=DEV) * 4.0 + 0.5
    kind = (torch.arange(M, device=DEV) % 2).to(torch.uint8)
    planes = ingest_bbox(bbox, kind, buffer=0.7)
    lists = ingest_bbox(bbox, kind, buffer=0.7, agent_major=True)
    assert lists.shape == (A, M, 2, 4) and torch.equal(planes_to_agents(planes), lists)
    with pytest.raises(ValueError):
        ingest_bbox(bbox, kind, tiled=True, agent_major=True)

    t = _case(A, M, 8, torch.float64)
    p_obs, p_ty = prepare_obs(t["obs"].clone(), t["obs_type"].clone())
    a_obs, a_ty = prepare_obs(planes_to_agents(t["obs"]), t["obs_type"].clone(), agents=A)
    assert torch.equal(planes_to_agents(p_obs), a_obs) and torch.equal(p_ty, a_ty)

    K = 9
    px = torch.linspace(0.0, 40.0, K, device=DEV)[:, None].repeat(1, A).contiguous()
    py = (0.02 * px * px + torch.rand((1, A), device=DEV)).contiguous()
    o_planes = torch.zeros((M, 2, A, 4), device=DEV)
    o_lists = torch.zeros((A, M, 2, 4), device=DEV)
    lane_fit(px, py, o_planes, 3, degree=2, buffer=0.3)
    lane_fit(px, py, o_lists, 3, degree=2, buffer=0.3)
    assert torch.equal(planes_to_agents(o_planes), o_lists)


def test_closed_loop_on_agent_major():
    """The rollout kernel (controller -> filter -> vehicle step, many ticks in one launch) reads per-agent lists."""
    from sccav_cbf_amd.batched import planes_to_agents
    from tests.test_gpu_nominal import _rollout_case
    make = _rollout_case(300, 16, 66, torch.float32, False)
    outs = []
    for conv in (None, planes_to_agents):
        loop, obs, _ = make()
        if conv:
            loop.obs = conv(obs)
        out = loop.rollout(20, motion="constant_velocity", trace=True)
        torch.cuda.synchronize()
        outs.append((loop.state.clone(), out["trace"].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_one_layout_per_call():
    import ctypes as C
    lib = _lib.load()
    flt = _filter()
    t = _case(64, 16, 1, torch.float32)
    u = torch.empty((64, 2), device=DEV)
    st = torch.empty((64,), dtype=torch.int32, device=DEV)
    for other in (_lib.FLAG_OBS_TILED, _lib.FLAG_OBS_SHARED):
        flt.params.flags = _lib.FLAG_OBS_AGENT_MAJOR | other
        rc = lib.sccbf_solve_f32(C.byref(flt.params), 64, 16, *[C.c_void_p(t[k].data_ptr()) for k in
                                                                 ("state", "u_ref", "obs", "obs_type")],
                                 C.c_void_p(u.data_ptr()), C.c_void_p(st.data_ptr()), None, None)
        assert rc == _lib.ERR_ARG
