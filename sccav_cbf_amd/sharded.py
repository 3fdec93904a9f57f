"""Agent-batch sharding over the GPUs of one node: one process per GPU.

Every QP is independent, so the solve has no exchange step: rank r owns the contiguous
agent block shard_bounds(A, world, r) and runs the filter on it.  Collectives
(torch.distributed: backend "nccl" = RCCL over xGMI on GPUs, "gloo" for CPU rehearsals)
appear only at the edges: scatter of a batch assembled on one rank, all-gather of the
controls.  gloo is only ever handed CPU tensors.

Two ways to get a rank its shard:
  local_batch   every rank makes its own agents from (seed, rank): no traffic at all (what a
                Monte-Carlo campaign and bench.py do)
  scatter       a batch assembled on one rank is cut into blocks and sent point to point, each block
                once and at its own size; with align=64 the cuts fall on tile boundaries, so a rank's
                slice of tiled obstacle slots (ceil(A/64), M, 2, 64, 4) is one contiguous block; so is
                its slice of per-agent lists (A, M, 2, 4), wherever the cut falls
"""
import torch
import torch.distributed as dist

TILE = 64


def shard_bounds(A, world, rank, align=1):
    """[lo, hi) of rank's contiguous block.  align=1: the first A % world ranks hold one more agent.
    align=64 (tiled obstacle slots): blocks of whole tiles, the first ranks hold one more tile and the
    last block ends at A."""
    units = (A + align - 1) // align
    base, rem = divmod(units, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return min(lo * align, A), min(hi * align, A)


def shard_batch(batch, world, rank):
    """Slice a full batch dict (state (A,4), u_ref (A,2), obs planes (M,2,A,4), tiles
    (ceil(A/64),M,2,64,4), per-agent lists (A,M,2,4) or shared (M,2,4), obs_type (M,) or (M,A)) down to the
    rank's block; tiled obstacle slots are cut on tile boundaries."""
    from .batched import obs_layout
    from ._lib import FLAG_OBS_AGENT_MAJOR
    A = batch["state"].shape[0]
    obs, ty = batch["obs"], batch["obs_type"]
    lo, hi = shard_bounds(A, world, rank, align=TILE if obs.dim() == 5 else 1)
    out = {"state": batch["state"][lo:hi].contiguous(), "u_ref": batch["u_ref"][lo:hi].contiguous()}
    if obs.dim() == 3:
        out["obs"] = obs
    elif obs.dim() == 5:
        out["obs"] = obs[(lo + TILE - 1) // TILE:(hi + TILE - 1) // TILE].contiguous()   # (an empty block: lo = hi = A)
    elif obs_layout(obs, A)[0] == FLAG_OBS_AGENT_MAJOR:
        out["obs"] = obs[lo:hi].contiguous()
    else:
        out["obs"] = obs[:, :, lo:hi, :].contiguous()
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

    def local_batch(self, make, agents_per_rank, seed):
        """This rank's own agents: make(A=agents_per_rank, seed=seed + rank) -> batch dict.  No
        communication; ranks differ by their seed only."""
        return make(A=agents_per_rank, seed=seed + self.rank)

    def scatter(self, batch, A, M, dtype, device, src=0, per_agent_types=False, tiled=False, agent_major=False):
        """Distribute a batch that lives on `src` (other ranks pass None); returns this rank's
        shard on `device`.  Every block travels once, at its own size (no padding), point to point.
        tiled: batch["obs"] is (ceil(A/64), M, 2, 64, 4) and the blocks are cut on tile boundaries.
        agent_major: batch["obs"] is (A, M, 2, 4)."""
        if tiled and agent_major:
            raise ValueError("one layout: tiled or agent_major")
        world, rank = self.world, self.rank
        align = TILE if tiled else 1
        cdev = torch.device("cpu") if _is_gloo(self.group) else device

        def parts(r):
            lo, hi = shard_bounds(A, world, r, align)
            n = hi - lo
            shapes = [("state", (n, 4), dtype), ("u_ref", (n, 2), dtype),
                      ("obs", ((n + TILE - 1) // TILE, M, 2, TILE, 4) if tiled else
                       ((n, M, 2, 4) if agent_major else (M, 2, n, 4)), dtype)]
            if per_agent_types:
                shapes.append(("obs_type", (M, n), torch.uint8))
            return lo, hi, shapes

        def cut(name, lo, hi):
            t = batch[name]
            if name == "obs":
                if agent_major:
                    return t[lo:hi]
                return t[(lo + TILE - 1) // TILE:(hi + TILE - 1) // TILE] if tiled else t[:, :, lo:hi, :]
            if name == "obs_type":
                return t[:, lo:hi]
            return t[lo:hi]

        out = {}
        if rank == src:
            pending = []
            for r in range(world):
                lo, hi, shapes = parts(r)
                for name, shape, _ in shapes:
                    piece = cut(name, lo, hi).contiguous()
                    if r == src:
                        out[name] = piece.to(device)
                    elif piece.numel():
                        pending.append(dist.isend(self._comm(piece), dst=r, group=self.group))
            for req in pending:
                req.wait()
        else:
            _, _, shapes = parts(rank)
            for name, shape, dt in shapes:
                buf = torch.empty(shape, dtype=dt, device=cdev)
                if buf.numel():
                    dist.recv(buf, src=src, group=self.group)
                out[name] = buf.to(device)
        if not per_agent_types:
            ty = self._comm(batch["obs_type"]).clone() if rank == src else torch.empty((M,), dtype=torch.uint8, device=cdev)
            dist.broadcast(ty, src=src, group=self.group)
            out["obs_type"] = ty.to(device)
        return out

    def solve_local(self, shard):
        return self.solve(shard["state"], shard["u_ref"], shard["obs"], shard["obs_type"])

    def all_gather(self, u_local, status_local, A, align=1):
        """Controls and status of the whole batch on every rank (align as in the scatter)."""
        world = self.world
        sizes = [b - a for a, b in (shard_bounds(A, world, r, align) for r in range(world))]
        chunk = max(sizes) if sizes else 0
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
        us = [ug[r * chunk: r * chunk + sizes[r]] for r in range(world)]
        ss = [sg[r * chunk: r * chunk + sizes[r]] for r in range(world)]
        return torch.cat(us).to(dev), torch.cat(ss).to(dev)
