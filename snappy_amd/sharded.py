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

    def __init__(self, build_dir, data_tar, rank, world, local_ranks=0, share_walk=None):
        """local_ranks: how many of the `world` ranks share THIS node's cores (0 = the launcher's LOCAL_WORLD_SIZE, else a
        guess from the visible GPUs): a rank plans host threads and fill threads for its share of them.
        share_walk: None = every rank walks the whole tree (snaphash_shard_plan); a dict(device=None | "cuda", group=None) =
        the ranks SHARE the walk (ABI 5 snaphash_shard_list / _plan_from): each walks the subtrees of every world-th entry
        of the root, two small all-gathers (lengths, then bytes) move the listings, every rank rebuilds the same records."""
        import ctypes
        h = ctypes.c_void_p()
        self.rank, self.world = rank, world
        self.rows = self.count = self.streams = self.bytes = 0
        self.fingerprint = 0
        self._hash_rc = 0
        self._h = None
        self._force = bool(share_walk and share_walk.get("force"))  # (a one-rank rehearsal of the collectives: bench.py's SNAPHASH_BENCH_FORCE_DIST)
        if share_walk is not None and (world > 1 or self._force):
            rc = self._plan_shared(build_dir, data_tar, share_walk.get("device"), share_walk.get("group"), h)
        else:
            rc = _lib.lib().snaphash_shard_plan(build_dir.encode(), data_tar.encode(), rank, world, ctypes.byref(h))
        self._h = h if not rc else None
        self._plan_rc = rc
        if rc:
            if world == 1 or rc == _lib.EINVAL:  # (a bad argument is the caller's bug, not a rank's misfortune)
                raise _lib.SnaphashError(rc, build_dir)
            return  # with other ranks about: reported at gather(), where every rank learns of it (see agree())
        L = _lib.lib()
        if local_ranks:
            L.snaphash_shard_set_local_ranks(h, local_ranks)
        self.rows = L.snaphash_shard_rows(h)
        self.count = L.snaphash_shard_count(h)
        self.streams = L.snaphash_shard_streams(h)
        self.bytes = L.snaphash_shard_bytes(h)
        self.fingerprint = L.snaphash_shard_fingerprint(h)

    def _plan_shared(self, build_dir, data_tar, device, group, h):
        """The shared walk: this rank's listing, the all-gather of the listings, the plan from all of them.  A rank whose
        listing failed says so in the first collective (a negative length) and EVERY rank raises: nobody is left waiting."""
        import ctypes
        import torch
        import torch.distributed as dist
        L = _lib.lib()
        blob, n = ctypes.c_void_p(), ctypes.c_size_t()
        rc = L.snaphash_shard_list(build_dir.encode(), self.rank, self.world, ctypes.byref(blob), ctypes.byref(n))
        try:
            dev = device or "cpu"
            mine = torch.tensor([rc if rc else n.value], dtype=torch.int64, device=dev)
            lens = torch.empty(self.world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(lens, mine, group=group)
            lens = lens.cpu().tolist()
            bad = [(r, v) for r, v in enumerate(lens) if v < 0]
            if bad:
                raise _lib.SnaphashError(int(bad[0][1]), "rank %d could not list its share of %s (every rank raises: nobody is left in the collective)" % (bad[0][0], build_dir))
            width = max(max(lens), 1)
            buf = np.zeros(width, dtype=np.uint8)
            if n.value:
                buf[:n.value] = np.ctypeslib.as_array(ctypes.cast(blob, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value,))
            t = torch.from_numpy(buf)
            if device is not None:
                t = t.to(device)
            every = torch.empty(self.world * width, dtype=torch.uint8, device=t.device)
            dist.all_gather_into_tensor(every, t, group=group)
            every = every.cpu().numpy()
        finally:
            if blob:
                L.snaphash_free(blob)
        ptrs = (ctypes.c_void_p * self.world)(*[every.ctypes.data + r * width for r in range(self.world)])
        sizes = (ctypes.c_size_t * self.world)(*[int(v) for v in lens])
        return L.snaphash_shard_plan_from(build_dir.encode(), data_tar.encode(), self.rank, self.world, ptrs, sizes, ctypes.byref(h))

    def paths(self):
        L = _lib.lib()
        return [L.snaphash_shard_path(self._h, k).decode() for k in range(self.count)]

    def hash(self, ctx, out=None):
        """Hashes this rank's members on ctx; returns the slab as a numpy [rows, 64] uint8 array (out: reuse one).
        With other ranks about, a failure here is not raised here: the rank must still reach the collective, where
        every rank learns of it (agree)."""
        if self._plan_rc:
            return np.zeros((max(self.rows, 1), 64), dtype=np.uint8)
        slab = out if out is not None else np.zeros((self.rows, 64), dtype=np.uint8)
        rc = _lib.lib().snaphash_shard_hash(ctx._h, self._h, slab.ctypes.data)
        if rc and self.world == 1:
            ctx._check(rc)
        self._hash_rc = rc
        self._hash_err = (_lib.lib().snaphash_last_error(ctx._h) or b"").decode(errors="replace") if rc else ""
        return slab

    def agree(self, device=None, group=None):
        """Before the all-gather: every rank's (plan rc, hash rc, streams, rows, fingerprint of the plan), all-gathered.
        Raises on EVERY rank if any rank failed or walked another tree -- the reference's serial loop returns its first
        error (snappy/build.go:242-244); a rank that left alone would leave the others blocked in the collective until
        the process group times out, and slabs of different plans would be gathered into the wrong rows.
        A caller that drives the C API directly owes the same check (snaphash_shard_fingerprint)."""
        import torch
        import torch.distributed as dist
        if self.world == 1 and not getattr(self, "_force", False):
            return
        fp = int(self.fingerprint)
        mine = torch.tensor([self._plan_rc, self._hash_rc, self.streams, self.rows, fp & 0x7fffffff, (fp >> 31) & 0x7fffffff, fp >> 62],
                            dtype=torch.int64, device=device or "cpu")
        every = torch.empty((self.world, mine.numel()), dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(every.view(-1), mine, group=group)
        every = every.cpu().tolist()
        for r, row in enumerate(every):
            if row[0] or row[1]:
                what = "snaphash_shard_plan" if row[0] else "snaphash_shard_hash"
                detail = (": " + self._hash_err) if r == self.rank and row[1] and getattr(self, "_hash_err", "") else ""
                raise _lib.SnaphashError(row[0] or row[1], "rank %d failed in %s%s (every rank raises: nobody is left in the collective)" % (r, what, detail))
        for r, row in enumerate(every):
            if row[2:] != every[0][2:]:
                raise _lib.SnaphashError(_lib.EMISMATCH, "rank %d walked another tree than rank 0 (%d streams / %d rows / plan %x against %d / %d / %x): "
                                         "the tree changed between the ranks' walks" % (r, row[2], row[3], row[4] | row[5] << 31 | row[6] << 62,
                                                                                          every[0][2], every[0][3], every[0][4] | every[0][5] << 31 | every[0][6] << 62))

    def gather(self, slab, device=None, group=None):
        """All ranks' slabs, rank-major [world * rows, 64], on the host.  device: where the collective runs ("cuda" for
        RCCL; None = CPU tensors over gloo).  The ranks first agree that they hold the same plan and that nobody failed
        (agree): 56 bytes a rank in front of the slabs."""
        import torch
        import torch.distributed as dist
        if self.world == 1 and not getattr(self, "_force", False):
            return slab
        self.agree(device=device, group=group)
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
