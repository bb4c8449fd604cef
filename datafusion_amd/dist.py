"""z-sharding helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The volume is split along z.  The ST term (u) is local to a shard; the only
exchanges per CG iteration are the sum of the spc-sized RD partial v and a handful of scalars.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def z_partition(n_z: int, world: int, rank: int) -> tuple[int, int]:
    """Planes [z0, z1) owned by `rank`: contiguous, sizes differ by at most one."""
    if world > n_z:
        raise ValueError(f"cannot shard {n_z} z-planes over {world} ranks")
    base, rem = divmod(n_z, world)
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


class ShardComm:
    """Sum-all-reduce over the shard group; the identity when there is one shard."""

    def __init__(self, group=None):
        if group is None or not dist.is_initialized():
            self.group, self.size, self.rank = None, 1, 0
            self._stage = False
        else:
            self.group = group if group is not True else dist.group.WORLD
            self.size = dist.get_world_size(self.group)
            self.rank = dist.get_rank(self.group)
            # a gloo group cannot reduce device memory: stage through the host (debug / test set-ups where
            # several ranks share one GPU; production groups are "nccl" = RCCL and reduce in place)
            self._stage = dist.get_backend(self.group) == "gloo"
        # DFX_FORCE_COMM=1: issue the collectives even on a one-rank group (a sum over one rank is the identity);
        # lets a single-GPU box exercise the real RCCL call path
        self._always = self.group is not None and os.environ.get("DFX_FORCE_COMM") == "1"

    def measure_allreduce(self, t: torch.Tensor, reps: int = 2):
        """Wall time (s) of one sum-all-reduce of `t` and of an 8-element one, as this group delivers them right now:
        max over ranks (every rank must reach the same slicing decision from it).  `t`'s contents are summed in place
        (pass scratch).  Used once per solve by FusionCG to size the overlap of the v all-reduce."""
        import time
        small = torch.zeros(8, dtype=t.dtype, device=t.device)
        out = []
        for buf in (t, small):
            self.sum_(buf)                                   # warm: communicator set-up, first-touch
            if buf.device.type == "cuda":
                torch.cuda.synchronize(buf.device)
            t0 = time.perf_counter()
            for _ in range(reps):
                self.sum_(buf)
            if buf.device.type == "cuda":
                torch.cuda.synchronize(buf.device)
            out.append((time.perf_counter() - t0) / reps)
        m = torch.tensor(out, dtype=torch.float64, device=t.device if not self._stage else "cpu")
        if self.size > 1:
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        return float(m[0]), float(m[1])

    def sum_async(self, t: torch.Tensor):
        """Start a sum-all-reduce of `t` ordered after the work already queued on the current stream; returns a
        handle for wait() (None when nothing is pending: one shard, or a host-staged gloo group)."""
        if not (self.size > 1 or self._always):
            return None
        if self._stage and t.device.type != "cpu":
            self.sum_(t)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @staticmethod
    def wait(handles) -> None:
        """Make the current stream wait for the collectives started by sum_async (no host block with RCCL)."""
        for h in handles:
            if h is not None:
                h.wait()

    def sum_(self, t: torch.Tensor) -> torch.Tensor:
        if self.size > 1 or self._always:
            if self._stage and t.device.type != "cpu":
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t
