"""Several batches in one launch (sccbf_solve_batches_*) against one sccbf_solve_* launch per batch: the same
bits, for every obstacle kind, ragged batch sizes, every layout and every kernel variant; and against the
fp64 oracle.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sccav_cbf_amd import _lib, scenarios  # noqa: E402

pytestmark = pytest.mark.gpu

R_HALF = np.diag([0.5, 0.5])
DEV = "cuda:0"
RAGGED = (1, 63, 64, 65, 0, 1000, 4097, 130)


def _filter(lanes=0, dynamics="dbm", slip_io=True, **kw):
    from sccav_cbf_amd.batched import BatchedCBF
    return BatchedCBF(alpha=1.0, R=R_HALF, lanes_per_qp=lanes, slip_io=slip_io, dynamics=dynamics, **kw)


def _case(A, M, seed, dtype, layout="planes", per_agent=False):
    """One batch with random slot kinds (ellipse, cone, lane, radial, empty, prepared, boxes) on the device."""
    from sccav_cbf_amd.batched import planes_to_tiled
    from tests.campaigns.fuzz_parity import random_case
    rng = np.random.default_rng(seed)
    state, u_ref, obs, kinds = random_case(rng, max(A, 1), M)
    state, u_ref, obs = state[:A], u_ref[:A], np.ascontiguousarray(obs[:, :, :A])
    ty = np.ascontiguousarray(np.broadcast_to(kinds[:, None], (M, A))) if per_agent else kinds
    host = dict(state=state, u_ref=u_ref, obs=obs, obs_type=kinds)
    t = dict(state=torch.from_numpy(state).to(DEV).to(dtype), u_ref=torch.from_numpy(u_ref).to(DEV).to(dtype),
             obs=torch.from_numpy(obs).to(DEV).to(dtype).contiguous(), obs_type=torch.from_numpy(ty).to(DEV))
    if layout == "tiles":
        t["obs"] = planes_to_tiled(t["obs"])
    return host, t


def _separate(flt, batches, want_rows=True):
    outs = []
    for t in batches:
        A = t["state"].shape[0]
        if A == 0:
            outs.append(None)
            continue
        outs.append(flt.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], want_rows=want_rows))
    torch.cuda.synchronize()
    return outs


def _assert_same(table, outs):
    for i, (b, o) in enumerate(zip(table.batches, outs)):
        if o is None:
            continue
        assert torch.equal(b["u_out"], o[0]), "u_out of batch %d" % i
        assert torch.equal(b["status"], o[1]), "status of batch %d" % i
        if len(o) > 2:
            assert torch.equal(b["rows_out"], o[2]), "rows_out of batch %d" % i


@pytest.mark.parametrize("lanes", (0, 1, 2, 4, 8, 16, 32, 64))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("layout", ["planes", "tiles"])
def test_fused_batches_equal_separate_launches(lanes, dtype, layout):
    """Ragged batches (an empty one among them), every slot kind, 16 rows and a row count beyond the registers of
    the narrow variants: u_out, status and rows_out bit for bit those of one launch per batch."""
    from sccav_cbf_amd.batched import BatchTable
    for M in (16, 37):
        if lanes not in (0, 32, 64) and M > 16 * lanes:
            continue
        flt = _filter(lanes=lanes)
        batches = [_case(A, M, 900 + 13 * i + M, dtype, layout, per_agent=(i % 3 == 2))[1]
                   for i, A in enumerate(RAGGED)]
        # one launch takes one layout of obs_type: per-agent kinds for all batches or for none
        for per_agent in (False, True):
            part = [b for b in batches if (b["obs_type"].dim() == 2) == per_agent]
            table = BatchTable(part, want_rows=True)
            for b in table.batches:
                b["u_out"].fill_(-7.0)
                b["status"].fill_(-7)
            flt.solve_batches(table)
            torch.cuda.synchronize()
            _assert_same(table, _separate(flt, part))


@pytest.mark.parametrize("M,lanes", [(129, 0), (200, 16), (300, 32), (513, 64), (900, 64), (1024, 0)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_fused_batches_with_rows_in_lds(M, lanes, dtype):
    """The kernels that keep the rows in LDS only (beyond 128 rows), paired workgroups included."""
    from sccav_cbf_amd.batched import BatchTable
    flt = _filter(lanes=lanes)
    batches = [_case(A, M, 77 + i, dtype, "tiles" if i % 2 else "planes")[1] for i, A in enumerate((5, 70, 9, 33))]
    for layout_dim in (4, 5):
        part = [b for b in batches if b["obs"].dim() == layout_dim]
        table = BatchTable(part, want_rows=True)
        flt.solve_batches(table)
        torch.cuda.synchronize()
        _assert_same(table, _separate(flt, part))


@pytest.mark.parametrize("dynamics", ["dbm", "kbm"])
def test_fused_batches_match_the_oracle(oracle, dynamics):
    """fp64 against the oracle (status identical, u to 1e-8), steering transforms included."""
    from sccav_cbf_amd.batched import BatchTable
    flt = _filter(dynamics=dynamics, slip_io=False)
    cases = [_case(A, 16, 31 + A, torch.float64) for A in (257, 1000, 64)]
    if dynamics == "kbm":
        for host, t in cases:
            host["u_ref"][:, 0] = np.abs(host["u_ref"][:, 0]) + 1.0      # a tracked speed, not an acceleration
            t["u_ref"].copy_(torch.from_numpy(host["u_ref"]))
    table = BatchTable([t for _, t in cases])
    flt.solve_batches(table)
    torch.cuda.synchronize()
    p = oracle.make_params(alpha=1.0, R=R_HALF.reshape(-1), dynamics=1 if dynamics == "kbm" else 0)
    for (host, _), b in zip(cases, table.batches):
        uo, so = oracle.solve_batch(p, host["state"], host["u_ref"], host["obs"], host["obs_type"])[:2]
        assert np.array_equal(b["status"].cpu().numpy(), so)
        ug = b["u_out"].cpu().numpy()
        ok = (so == 0) & (np.abs(uo[:, 1]) < 1.2)
        assert (np.abs(ug - uo)[ok] / np.maximum(1.0, np.abs(uo[ok]))).max() <= 1e-8
        assert np.array_equal(ug[so != 0], uo[so != 0])


def test_records_the_kernel_cannot_serve_are_skipped():
    """A record without agents, with a pointer missing or with more agents than A_max is left alone: none of its
    outputs is written; the other batches of the launch are solved."""
    from sccav_cbf_amd.batched import BatchTable
    lib = _lib.load()
    for M, lanes in ((16, 4), (300, 32), (0, 0)):
        flt = _filter(lanes=lanes)
        batches = [_case(A, max(M, 1), 5 + A, torch.float32)[1] for A in (700, 64, 3000, 129)]
        if M == 0:
            for b in batches:
                b["obs"], b["obs_type"] = b["obs"][:0], b["obs_type"][:0]
        table = BatchTable(batches)
        want = _separate(flt, batches, want_rows=False)
        rec = table.host_records.copy()
        if M:
            rec[1, 4] = 0                       # obs_type missing
        else:
            rec[1, 5] = 0                       # u_out missing
        rec[3, 0] = 1 << 31                     # more agents than the kernels index
        dev_rec = torch.from_numpy(rec).to(DEV)
        for a_max, skipped in ((3000, (1, 3)), (700, (1, 2, 3))):
            for b in table.batches:
                b["u_out"].fill_(-7.0)
                b["status"].fill_(-7)
            flt.params.flags = _lib.FLAG_SLIP_IO
            _lib.check(lib.sccbf_solve_batches_f32(C.byref(flt.params), 4, M, a_max, C.c_void_p(dev_rec.data_ptr()),
                                                   None))
            torch.cuda.synchronize()
            for i, (b, o) in enumerate(zip(table.batches, want)):
                if i in skipped:
                    assert (b["u_out"] == -7.0).all() and (b["status"] == -7).all(), (M, a_max, i)
                else:
                    assert torch.equal(b["u_out"], o[0]) and torch.equal(b["status"], o[1]), (M, a_max, i)


def test_shared_obstacle_table_per_batch():
    """SCCBF_FLAG_OBS_SHARED: every batch brings one table for all its agents."""
    from sccav_cbf_amd.batched import BatchTable
    flt = _filter()
    batches = []
    for i, A in enumerate((100, 1, 333)):
        _, t = _case(A, 12, 400 + i, torch.float64)
        t["obs"] = t["obs"][:, :, 0, :].contiguous()     # the first agent's obstacles, for everybody
        batches.append(t)
    table = BatchTable(batches, want_rows=True)
    flt.solve_batches(table)
    torch.cuda.synchronize()
    _assert_same(table, _separate(flt, batches))


def test_part_of_a_table_and_repeated_launches():
    """A launch may take a slice of the records; launching again overwrites the same outputs with the same bits."""
    from sccav_cbf_amd.batched import BatchTable
    flt = _filter(streamed_obs=True)
    batches = [_case(512, 16, 60 + i, torch.float32, "tiles")[1] for i in range(6)]
    table = BatchTable(batches)
    want = _separate(flt, batches, want_rows=False)
    for b in table.batches:
        b["u_out"].fill_(-7.0)
    flt.solve_batches(table, first=2, count=3)
    torch.cuda.synchronize()
    for i, (b, o) in enumerate(zip(table.batches, want)):
        if 2 <= i < 5:
            assert torch.equal(b["u_out"], o[0]) and torch.equal(b["status"], o[1])
        else:
            assert (b["u_out"] == -7.0).all()
    for _ in range(3):
        flt.solve_batches(table)
    torch.cuda.synchronize()
    _assert_same(table, want)


def test_full_size_batches():
    """Sixteen batches of the bench workload (65536 agents x 16 rows, tiled prepared slots, streamed hint): the
    answers of sixteen launches, bit for bit."""
    from sccav_cbf_amd.batched import BatchTable, planes_to_tiled
    flt = _filter(slip_io=False, streamed_obs=True)
    w = scenarios.prepared(scenarios.config4(A=65536, nominal="uniform"))
    batches = []
    for i in range(16):
        sh = 64 * 131 * i
        batches.append(dict(
            state=torch.from_numpy(np.roll(w["state"], sh, axis=0)).float().to(DEV),
            u_ref=torch.from_numpy(np.roll(w["u_ref"], sh, axis=0)).float().to(DEV),
            obs=planes_to_tiled(torch.from_numpy(np.roll(w["obs"], sh, axis=2)).float().to(DEV)),
            obs_type=torch.from_numpy(w["obs_type"]).to(DEV)))
    table = BatchTable(batches)
    flt.solve_batches(table)
    torch.cuda.synchronize()
    _assert_same(table, _separate(flt, batches, want_rows=False))
    # rotating the agents rotates the answers
    u0 = table.batches[0]["u_out"]
    assert torch.equal(torch.roll(u0, 64 * 131 * 5, dims=0), table.batches[5]["u_out"])
