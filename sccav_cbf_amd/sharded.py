"""Agent-batch sharding over the GPUs of one node: one process per GPU.

Every QP is independent, so the solve has no exchange step: rank r owns the contiguous
agent block shard_bounds(A, world, r) and runs the filter on it.  Collectives
(torch.distributed: backend "nccl" = RCCL over xGMI on GPUs, "gloo" for CPU rehearsals)
appear only at the edges: scatter of a batch assembled on one rank, all-gather of the
controls.  gloo is only ever handed CPU tensors.
"""
import torch
import torch.distributed as dist


def shard_bounds(A, world, rank):
    """[lo, hi) of rank's contiguous block; the first A % world ranks hold one more agent."""
    base, rem = divmod(A, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch, world, rank):
    """Slice a full batch dict (state (A,4), u_ref (A,2), obs (M,2,A,4) or shared (M,2,4),
    obs_type (M,) or (M,A)) down to the rank's block."""
    A = batch["state"].shape[0]
    lo, hi = shard_bounds(A, world, rank)
    out = {"state": batch["state"][lo:hi].contiguous(), "u_ref": batch["u_ref"][lo:hi].contiguous()}
    obs, ty = batch["obs"], batch["obs_type"]
    out["obs"] = obs if obs.dim() == 3 else obs[:, :, lo:hi, :].contiguous()
    out["obs_type"] = ty if ty.dim() == 1 else ty[:, lo:hi].contiguous()
    return out


def _is_gloo(group):
    return dist.get_backend(group) == "gloo"


class ShardedCBF:
    """Runs `solve(state, u_ref, obs, obs_type) -> (u, status)` on this rank's shard and moves
    batches / results between ranks.  `solve` is normally BatchedCBF.solve."""

    def __init__(self, solve, group=None):
        self.solve = solve
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def _comm(self, t):
        """Tensor as the backend wants it: host memory for gloo (synchronous copy)."""
        return t.cpu() if (_is_gloo(self.group) and t.is_cuda) else t

    def scatter(self, batch, A, M, dtype, device, src=0, per_agent_types=False):
        """Distribute a batch that lives on `src` (other ranks pass None); returns this rank's
        shard on `device`.  Blocks are padded to equal length for the collective."""
        world, rank = self.world, self.rank
        chunk = (A + world - 1) // world
        lo, hi = shard_bounds(A, world, rank)
        n = hi - lo
        cdev = torch.device("cpu") if _is_gloo(self.group) else device

        def scatter_rows(full, tail_shape, dt, agent_dim):
            # full: tensor with the agent axis at agent_dim (src only)
            recv = torch.empty((chunk,) + tail_shape, dtype=dt, device=cdev)
            parts = None
            if rank == src:
                moved = self._comm(full).movedim(agent_dim, 0)
                parts = []
                for r in range(world):
                    a, b = shard_bounds(A, world, r)
                    p = torch.zeros((chunk,) + tail_shape, dtype=dt, device=cdev)
                    p[: b - a] = moved[a:b]
                    parts.append(p)
            dist.scatter(recv, parts, src=src, group=self.group)
            return recv[:n].movedim(0, agent_dim).contiguous().to(device)

        out = {"state": scatter_rows(batch["state"] if rank == src else None, (4,), dtype, 0),
               "u_ref": scatter_rows(batch["u_ref"] if rank == src else None, (2,), dtype, 0),
               "obs": scatter_rows(batch["obs"] if rank == src else None, (M, 2, 4), dtype, 2)}
        if per_agent_types:
            out["obs_type"] = scatter_rows(batch["obs_type"] if rank == src else None, (M,), torch.uint8, 1)
        else:
            ty = self._comm(batch["obs_type"]).clone() if rank == src else torch.empty((M,), dtype=torch.uint8, device=cdev)
            dist.broadcast(ty, src=src, group=self.group)
            out["obs_type"] = ty.to(device)
        return out

    def solve_local(self, shard):
        return self.solve(shard["state"], shard["u_ref"], shard["obs"], shard["obs_type"])

    def all_gather(self, u_local, status_local, A):
        """Controls and status of the whole batch on every rank."""
        world = self.world
        chunk = (A + world - 1) // world
        dev = u_local.device
        n = u_local.shape[0]
        up = torch.zeros((chunk, 2), dtype=u_local.dtype, device=dev)
        sp = torch.zeros((chunk,), dtype=status_local.dtype, device=dev)
        up[:n], sp[:n] = u_local, status_local
        up, sp = self._comm(up), self._comm(sp)
        ug = torch.empty((world * chunk, 2), dtype=up.dtype, device=up.device)
        sg = torch.empty((world * chunk,), dtype=sp.dtype, device=sp.device)
        dist.all_gather_into_tensor(ug, up, group=self.group)
        dist.all_gather_into_tensor(sg, sp, group=self.group)
        us, ss = [], []
        for r in range(world):
            a, b = shard_bounds(A, world, r)
            us.append(ug[r * chunk: r * chunk + (b - a)])
            ss.append(sg[r * chunk: r * chunk + (b - a)])
        return torch.cat(us).to(dev), torch.cat(ss).to(dev)
