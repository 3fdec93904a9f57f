"""One process per rank on the GPU: two ranks (sharing the one card of the test box) scatter a batch,
run the HIP filter on their shards and all-gather the controls; gloo only ever sees host tensors.
The gathered result must be bit-identical to one unsharded solve."""
import os
import socket
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from sccav_cbf_amd import scenarios  # noqa: E402
from sccav_cbf_amd.sharded import ShardedCBF, shard_bounds  # noqa: E402

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, A, tiled, ret, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from sccav_cbf_amd.batched import BatchedCBF
        flt = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]))
        M = 16
        batch = None
        if rank == 0:
            w = scenarios.config4(A=A, nominal="uniform")
            batch = {k: torch.from_numpy(v) for k, v in w.items()}
            batch = {k: (v.float() if v.dtype.is_floating_point else v) for k, v in batch.items()}
            if backend == "nccl":          # RCCL moves device memory
                batch = {k: v.to(dev) for k, v in batch.items()}
        from sccav_cbf_amd.batched import planes_to_tiled
        planes = None
        if rank == 0 and tiled:      # the layout the benchmark uses: a rank's tiles are one contiguous block
            planes = batch["obs"]
            batch["obs"] = planes_to_tiled(planes)
        sh = ShardedCBF(flt.solve)
        align = 64 if tiled else 1
        shard = sh.scatter(batch, A, M, torch.float32, dev, tiled=tiled)
        lo, hi = shard_bounds(A, world, rank, align)
        assert shard["state"].is_cuda
        assert shard["obs"].shape == (((hi - lo + 63) // 64, M, 2, 64, 4) if tiled else (M, 2, hi - lo, 4))
        u, st = sh.solve_local(shard)
        torch.cuda.synchronize()
        ug, sg = sh.all_gather(u, st, A, align=align)
        if rank == 0:
            if tiled:
                batch["obs"] = planes
            full = {k: v.to(dev) for k, v in batch.items()}
            uf, sf = flt.solve(full["state"], full["u_ref"], full["obs"], full["obs_type"])
            torch.cuda.synchronize()
            ret["equal"] = bool(torch.equal(ug, uf) and torch.equal(sg, sf))
            ret["n"] = int(ug.shape[0])
            ret["active"] = int((sf == 0).sum())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tiled", [False, True], ids=["planes", "tiles"])
def test_two_ranks_on_the_gpu_scatter_solve_gather(tiled):
    world, A = 2, 5001
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.spawn(_worker, args=(world, _free_port(), A, tiled, ret), nprocs=world, join=False)
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > 150:
            for p in ctx.processes:
                p.terminate()
            pytest.fail("ranks did not finish in 150 s")
    assert ret["n"] == A and ret["equal"] and ret["active"] > 1000


def test_one_rank_over_rccl_scatter_solve_gather():
    """The same path with the collectives on RCCL (backend "nccl", device tensors, no host copies): one rank, which is
    all a one-GPU box can start; broadcast and all-gather run through the library all the same."""
    A = 5001
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.spawn(_worker, args=(1, _free_port(), A, True, ret, "nccl"), nprocs=1, join=False)
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > 150:
            for p in ctx.processes:
                p.terminate()
            pytest.fail("the rank did not finish in 150 s")
    assert ret["n"] == A and ret["equal"] and ret["active"] > 1000
