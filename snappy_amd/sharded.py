"""Sharding one writeHashes file list over the GPUs of a node (one process per
GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Files are independent streams, so the only exchange step is the gather of the
digest vector at the end (64 B per file): every rank derives the same
deterministic LPT plan from the size list, hashes its own files into a
fixed-size slab (kmax rows), one all_gather_into_tensor moves the slabs and an
index_select puts the rows back into walk order.  No collective touches file
bytes.  A single file cannot be split across GPUs (Merkle-Damgard chaining).
"""
import numpy as np

from . import _lib


class ShardPlan:
    def __init__(self, sizes, world):
        self.sizes = np.ascontiguousarray(sizes, dtype=np.uint64)
        self.world = int(world)
        self.shard_of = _lib.lpt_assign(self.sizes, self.world)       # identical on every rank
        self.counts = np.bincount(self.shard_of, minlength=self.world)
        self.kmax = int(self.counts.max()) if len(self.sizes) else 0
        # row of the gathered [world*kmax, 64] slab matrix that holds global file i
        self.row_of = np.zeros(len(self.sizes), dtype=np.int64)
        self._members = []
        for r in range(self.world):
            idx = np.nonzero(self.shard_of == r)[0]
            self._members.append(idx)
            self.row_of[idx] = r * self.kmax + np.arange(len(idx))

    def members(self, rank):
        """Global file indices hashed by `rank`, in walk order."""
        return self._members[rank]


def gather_digests(local_slab, plan, group=None, force_collective=False):
    """local_slab: uint8 tensor [kmax, 64] (rows beyond this rank's count are
    ignored).  Returns the full [n_files, 64] digest matrix in walk order on every
    rank.  Works on CUDA tensors over RCCL and on CPU tensors over gloo."""
    import torch
    import torch.distributed as dist
    if plan.world == 1 and not force_collective:
        return local_slab[:len(plan.sizes)]
    gathered = torch.empty((plan.world * plan.kmax, 64), dtype=torch.uint8, device=local_slab.device)
    dist.all_gather_into_tensor(gathered.view(-1), local_slab.contiguous().view(-1), group=group)
    index = torch.from_numpy(plan.row_of).to(local_slab.device)
    return gathered.index_select(0, index)
