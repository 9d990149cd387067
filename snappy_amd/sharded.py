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


class ShardedTree:
    """writeHashes (snappy/build.go:216-270) with one process per GPU, natively (ABI 4 snaphash_shard_*): every rank
    walks the same tree and derives the same LPT plan, hashes ITS members into a slab, ONE all-gather (RCCL on CUDA
    tensors, gloo on CPU tensors) moves the slabs, rank 0 -- or every rank -- writes hashes.yaml.

        st = ShardedTree(build_dir, data_tar, rank, world)
        slab = st.hash(ctx)                  # this rank's digests, [rows, 64] uint8 (host)
        yaml = st.emit(st.gather(slab))      # all ranks' slabs -> hashes.yaml
    """

    def __init__(self, build_dir, data_tar, rank, world):
        import ctypes
        h = ctypes.c_void_p()
        rc = _lib.lib().snaphash_shard_plan(build_dir.encode(), data_tar.encode(), rank, world, ctypes.byref(h))
        if rc:
            raise _lib.SnaphashError(rc, build_dir)
        self._h = h
        self.rank, self.world = rank, world
        L = _lib.lib()
        self.rows = L.snaphash_shard_rows(h)
        self.count = L.snaphash_shard_count(h)
        self.streams = L.snaphash_shard_streams(h)
        self.bytes = L.snaphash_shard_bytes(h)

    def paths(self):
        L = _lib.lib()
        return [L.snaphash_shard_path(self._h, k).decode() for k in range(self.count)]

    def hash(self, ctx, out=None):
        """Hashes this rank's members on ctx; returns the slab as a numpy [rows, 64] uint8 array (out: reuse one)."""
        slab = out if out is not None else np.zeros((self.rows, 64), dtype=np.uint8)
        ctx._check(_lib.lib().snaphash_shard_hash(ctx._h, self._h, slab.ctypes.data))
        return slab

    def gather(self, slab, device=None, group=None):
        """All ranks' slabs, rank-major [world * rows, 64], on the host.  device: where the collective runs ("cuda" for
        RCCL; None = CPU tensors over gloo)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return slab
        t = torch.from_numpy(slab)
        if device is not None:
            t = t.to(device, non_blocking=False)
        full = torch.empty((self.world * self.rows, 64), dtype=torch.uint8, device=t.device)
        dist.all_gather_into_tensor(full.view(-1), t.contiguous().view(-1), group=group)
        return full.cpu().numpy()

    def emit(self, slabs):
        import ctypes
        slabs = np.ascontiguousarray(slabs, dtype=np.uint8)
        assert slabs.size == self.world * self.rows * 64
        out, n = ctypes.c_void_p(), ctypes.c_size_t()
        rc = _lib.lib().snaphash_shard_emit(self._h, slabs.ctypes.data, ctypes.byref(out), ctypes.byref(n))
        if rc:
            raise _lib.SnaphashError(rc)
        try:
            return ctypes.string_at(out, n.value)
        finally:
            _lib.lib().snaphash_free(out)

    def close(self):
        if self._h:
            _lib.lib().snaphash_shard_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
