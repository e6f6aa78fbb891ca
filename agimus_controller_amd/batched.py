"""Batch axis and multi-GPU sharding.

The reference runs one MPC per process; the batch of independent MPC instances is this
package's addition.  Instances never interact, so a node with G GPUs runs G processes (one per
GPU, `torch.distributed` with the RCCL backend) that each own a contiguous slice of the batch.
The only communication is the scatter of the inputs from / gather of the results to a root
rank -- `scatter_rows` / `gather_rows` below, point-to-point-sized payloads over xGMI.
"""

from __future__ import annotations

import numpy as np


def shard_bounds(n_instances: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of the batch owned by `rank` (first ranks get the remainder)."""
    base, rem = divmod(n_instances, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def scatter_rows(full, n_instances: int, root: int = 0, group=None, device="cpu"):
    """Root holds `full` [n_instances, ...]; every rank receives its slice (numpy in, numpy out)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = [None, None]  # trailing shape and dtype travel from the root
    if rank == root:
        full = np.ascontiguousarray(full)
        meta = [tuple(full.shape[1:]), str(full.dtype)]
    dist.broadcast_object_list(meta, src=root, group=group)
    tail, dtype = meta
    lo, hi = shard_bounds(n_instances, rank, world)
    out = torch.empty((hi - lo,) + tuple(tail), dtype=getattr(torch, dtype), device=device)
    if rank == root:
        pieces = []
        for r in range(world):
            a, b = shard_bounds(n_instances, r, world)
            pieces.append(torch.from_numpy(full[a:b]).to(device))
        out.copy_(pieces[root])
        reqs = [dist.isend(pieces[r], dst=r, group=group) for r in range(world) if r != root]
        for q in reqs:
            q.wait()
    else:
        dist.recv(out, src=root, group=group)
    return out.cpu().numpy()


def gather_rows(local, n_instances: int, root: int = 0, group=None, device="cpu"):
    """Inverse of scatter_rows: root returns the concatenation in instance order, others None."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    if rank != root:
        dist.send(t, dst=root, group=group)
        return None
    out = np.empty((n_instances,) + tuple(local.shape[1:]), dtype=local.dtype)
    for r in range(world):
        a, b = shard_bounds(n_instances, r, world)
        if r == root:
            out[a:b] = local
        else:
            buf = torch.empty((b - a,) + tuple(local.shape[1:]), dtype=t.dtype, device=device)
            dist.recv(buf, src=r, group=group)
            out[a:b] = buf.cpu().numpy()
    return out


class BatchedMPC:
    """B independent receding-horizon controllers resident on one GPU.

    `solver` is a backend.HipOcp (or any object with the same methods).  References live in HBM
    (`sine_trajectory` / `set_refs`), the previous solution never leaves the device; per step only
    what the controller publishes comes back (us[0], K[0], x1, status)."""

    def __init__(self, solver, max_iter: int):
        self.solver = solver
        self.max_iter = int(max_iter)
        self.k = 0

    def step(self):
        self.solver.mpc_step(self.k, self.max_iter, first=(self.k == 0))
        self.k += 1
        return self.solver.download_first()
