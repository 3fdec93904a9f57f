"""The N > 1 path rehearsed on CPU: two gloo ranks scatter a batch, solve their shards and
all-gather the controls; the result must be bit-identical to the unsharded solve.  The
per-shard solver here is the oracle (this test checks partitioning and data movement, the
kernels are checked in the gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sccav_cbf_amd import scenarios
from sccav_cbf_amd.sharded import ShardedCBF, shard_batch, shard_bounds


def test_shard_bounds_cover_and_balance():
    for A in (0, 1, 7, 8, 65536, 65537, 100003):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(A, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == A
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_tile_aligned_bounds():
    for A in (1, 63, 64, 65, 1000, 65536, 65537):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(A, world, r, align=64) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == A
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all(lo % 64 == 0 for lo, _ in b if lo < A)
            tiles = [(hi - lo + 63) // 64 for lo, hi in b]
            assert max(tiles) - min(tiles) <= 1


def test_shard_batch_slices():
    w = scenarios.config4(A=101, nominal="uniform")
    t = {k: torch.from_numpy(v) for k, v in w.items()}
    parts = [shard_batch(t, 3, r) for r in range(3)]
    assert torch.equal(torch.cat([p["state"] for p in parts]), t["state"])
    assert torch.equal(torch.cat([p["obs"] for p in parts], dim=2), t["obs"])
    assert all(p["obs"].is_contiguous() and p["obs_type"].shape == (16,) for p in parts)
    t["obs_type"] = t["obs_type"][:, None].repeat(1, 101)
    parts = [shard_batch(t, 3, r) for r in range(3)]
    assert torch.equal(torch.cat([p["obs_type"] for p in parts], dim=1), t["obs_type"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, A, per_agent, tiled, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        p = orc.make_params(alpha=1.0, R=(0.5, 0, 0, 0.5),
                            flags=orc.FLAG_TYPE_PER_AGENT if per_agent else 0)

        def solve(state, u_ref, obs, ty):
            u, st, _ = orc.solve_batch(p, state.numpy(), u_ref.numpy(), obs.numpy(), ty.numpy())
            return torch.from_numpy(u), torch.from_numpy(st)

        M = 16
        batch = None
        if rank == 0:
            w = scenarios.config4(A=A, nominal="uniform")
            if per_agent:
                ty = np.repeat(w["obs_type"][:, None], A, axis=1)
                ty[::3, ::5] = 255
                w["obs_type"] = np.ascontiguousarray(ty)
            batch = {k: torch.from_numpy(v) for k, v in w.items()}
        from sccav_cbf_amd.batched import planes_to_agents, planes_to_tiled
        lists = tiled == "lists"       # per-agent lists (A, M, 2, 4): cut anywhere, a rank's share is one block
        tiled = tiled is True
        planes = None
        if rank == 0:
            planes = batch["obs"]
            if tiled:
                batch["obs"] = planes_to_tiled(planes)
            elif lists:
                batch["obs"] = planes_to_agents(planes)
        sh = ShardedCBF(solve)
        align = 64 if tiled else 1
        shard = sh.scatter(batch, A, M, torch.float64, torch.device("cpu"), per_agent_types=per_agent, tiled=tiled,
                           agent_major=lists)
        lo, hi = shard_bounds(A, world, rank, align)
        assert shard["state"].shape == (hi - lo, 4)
        if lists:
            assert shard["obs"].shape == (hi - lo, M, 2, 4) and shard["obs"].is_contiguous()
            if rank == 0:
                from sccav_cbf_amd.sharded import shard_batch
                assert torch.equal(shard_batch(batch, world, rank)["obs"], shard["obs"])
            shard["obs"] = shard["obs"].permute(1, 2, 0, 3).contiguous()
        if tiled:    # a rank's tiles are one contiguous block of the batch; back to planes for the oracle
            assert shard["obs"].shape == ((hi - lo + 63) // 64, M, 2, 64, 4) and shard["obs"].is_contiguous()
            shard["obs"] = shard["obs"].permute(1, 2, 0, 3, 4).reshape(M, 2, -1, 4)[:, :, :hi - lo].contiguous()
        assert shard["obs"].shape == (M, 2, hi - lo, 4)
        u, st = sh.solve_local(shard)
        ug, sg = sh.all_gather(u, st, A, align=align)
        if rank == 0:
            batch["obs"] = planes
        if rank == 0:
            uf, sf = solve(batch["state"], batch["u_ref"], batch["obs"], batch["obs_type"])
            ret["equal"] = bool(torch.equal(ug, uf) and torch.equal(sg, sf))
            ret["n"] = int(ug.shape[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("A,per_agent,tiled", [(1000, False, False), (1001, True, False), (1001, False, True),
                                              (130, True, True), (40, False, True), (1001, False, "lists"),
                                              (131, True, "lists")])
def test_two_rank_gloo_scatter_solve_gather(A, per_agent, tiled):
    from oracle import oracle as orc
    orc.build()
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), A, per_agent, tiled, ret), nprocs=world, join=True)
    assert ret["n"] == A and ret["equal"]


def test_local_batches_differ_by_rank_only():
    """Every rank makes its own agents from (seed, rank): the no-traffic alternative to the scatter."""
    class _Fake(ShardedCBF):
        def __init__(self, rank):
            self.rank, self.world, self.group, self.solve = rank, 2, None, None
    a = _Fake(0).local_batch(lambda A, seed: scenarios.config4(A=A, seed=seed, nominal="uniform"), 50, 9)
    b = _Fake(1).local_batch(lambda A, seed: scenarios.config4(A=A, seed=seed, nominal="uniform"), 50, 9)
    a2 = _Fake(0).local_batch(lambda A, seed: scenarios.config4(A=A, seed=seed, nominal="uniform"), 50, 9)
    assert a["state"].shape == (50, 4) and not np.array_equal(a["state"], b["state"])
    assert np.array_equal(a["state"], a2["state"]) and np.array_equal(a["obs"], a2["obs"])
