"""sccbf_solve_batches_static, the entry point that allocates nothing (the records of the batches go to a table
the library itself holds in device memory), against sccbf_solve_batches_* on a table of the caller's: the same
bits, on two calls in a row that share the library's table, the second one smaller.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sccav_cbf_amd import _lib  # noqa: E402
from tests.test_gpu_batches import _case, _filter  # noqa: E402

pytestmark = pytest.mark.gpu


def _fresh(table):
    for b in table.batches:
        b["u_out"].fill_(-7.0)
        b["status"].fill_(-7)
        if b.get("rows_out") is not None:
            b["rows_out"].fill_(-7.0)


def _answers(table):
    return [(b["u_out"].clone(), b["status"].clone(), b["rows_out"].clone()) for b in table.batches]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("M,lanes", [(16, 0), (37, 8), (300, 0)])
def test_static_table_entry_equals_the_callers_table_entry(dtype, M, lanes):
    from sccav_cbf_amd.batched import BatchTable
    flt = _filter(lanes=lanes)
    # 70 batches: more than one fill launch (32 records each); then 3 smaller ones
    first = BatchTable([_case(A, M, 11 + i, dtype, "tiles" if M == 16 else "planes")[1]
                        for i, A in enumerate([97, 64, 1, 300, 0] * 14)], want_rows=True)
    second = BatchTable([_case(A, M, 500 + i, dtype, "tiles" if M == 16 else "planes")[1]
                         for i, A in enumerate([33, 150, 64])], want_rows=True)
    want = []
    for t in (first, second):
        _fresh(t)
        flt.solve_batches(t)
        torch.cuda.synchronize()
        want.append(_answers(t))
        _fresh(t)
    before = torch.cuda.memory_allocated()
    flt.solve_batches_static(first)       # two calls in a row on one stream: the second overwrites the head of the
    flt.solve_batches_static(second)      # library's table while nothing of the first is still to be read
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    for t, w in zip((first, second), want):
        for i, (b, (u, st, rows)) in enumerate(zip(t.batches, w)):
            assert torch.equal(b["u_out"], u), i
            assert torch.equal(b["status"], st), i
            assert torch.equal(b["rows_out"], rows), i
    # and once more the other way round, a part of a table included
    _fresh(first)
    _fresh(second)
    flt.solve_batches_static(second, first=1, count=2)
    flt.solve_batches_static(first, first=3, count=40)
    torch.cuda.synchronize()
    for t, w, lo, hi in ((second, want[1], 1, 3), (first, want[0], 3, 43)):
        for i, (b, (u, st, rows)) in enumerate(zip(t.batches, w)):
            if lo <= i < hi:
                assert torch.equal(b["u_out"], u) and torch.equal(b["status"], st) and torch.equal(b["rows_out"], rows)
            else:
                assert (b["status"] == -7).all()


def test_static_table_entry_refuses_what_it_cannot_hold():
    lib = _lib.load()
    p = _lib.Params()
    lib.sccbf_default_params(C.byref(p))
    recs = (_lib.Batch * (_lib.MAX_STATIC_BATCHES + 1))()
    assert lib.sccbf_solve_batches_static(C.byref(p), 4, _lib.MAX_STATIC_BATCHES + 1, 16, recs, None) == _lib.ERR_BATCHES
    assert b"table" in lib.sccbf_strerror(_lib.ERR_BATCHES)


def test_static_table_at_its_limit():
    """1024 batches in one call: every record of the library's table in use."""
    from sccav_cbf_amd.batched import BatchTable
    flt = _filter()
    _, t = _case(64, 16, 3, torch.float32)
    K = _lib.MAX_STATIC_BATCHES
    batches = [dict(state=t["state"], u_ref=t["u_ref"], obs=t["obs"], obs_type=t["obs_type"]) for _ in range(K)]
    table = BatchTable(batches)
    u, st = flt.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"])
    _fresh(table)
    flt.solve_batches_static(table)
    torch.cuda.synchronize()
    for b in table.batches:
        assert torch.equal(b["u_out"], u) and torch.equal(b["status"], st)
