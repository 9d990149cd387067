"""The sharded pass end to end with real hashing: two ranks (processes) share the one GPU of the
test box, each hashes its LPT shard of a Zipf-sized file list through the C ABI, the digest slabs
are gathered (gloo here, because RCCL refuses two ranks on one device; bench.py uses RCCL with one
GPU per rank) and every rank must hold the full digest vector in walk order, bit-exact."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sizes, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snappy_amd import Context, synthetic
        from snappy_amd.sharded import ShardPlan, gather_digests
        plan = ShardPlan(sizes, world)
        mine = plan.members(rank)
        lens = np.ascontiguousarray(sizes[mine])
        off, total = synthetic.pack_offsets(lens)
        with Context(device=0) as ctx:
            data = torch.empty(max(total, 16), dtype=torch.uint8, device="cuda")
            ctx.fill_synthetic_device(data.data_ptr(), off, lens, np.ascontiguousarray(mine.astype(np.uint64)))
            slab = torch.zeros((plan.kmax, 64), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # the ctx stream is non-blocking: order it after torch's memset
            ctx.sha512_device(data.data_ptr(), off, lens, slab.data_ptr())
            ctx.sync()
            full = gather_digests(slab.cpu(), plan)
        q.put((rank, full.numpy().tobytes(), int(len(mine))))
    finally:
        dist.destroy_process_group()


@pytest.mark.kernels_only("the HBM-resident entry point never plans")
def test_two_ranks_shard_hash_gather(oracle):
    import torch.multiprocessing as mp
    from snappy_amd import synthetic
    sizes = np.minimum(synthetic.zipf_sizes(3000), np.uint64(1 << 22))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = b"".join(oracle.sha512(oracle.fill_synthetic(int(n), i).tobytes()) for i, n in enumerate(sizes))
    assert sum(r[2] for r in res) == len(sizes)
    for rank, blob, _ in res:
        assert blob == want, rank


def _tree_worker(rank, world, port, build, tar, q, planned):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch  # noqa: F401
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snappy_amd import Context, _lib
        from snappy_amd.sharded import ShardedTree
        with Context(device=0, flags=0 if planned else _lib.FLAG_GPU_ONLY) as ctx, ShardedTree(build, tar, rank, world, local_ranks=world, share_walk={} if planned else None) as st:  # (planned: the ranks also share the walk, ABI 5)
            slab = st.hash(ctx)
            ex = ctx.stats_ex()
            y = st.emit(st.gather(slab))
        q.put((rank, y, st.count, int(ex["gpu_bytes"]), int(ex["host_bytes"])))
    finally:
        dist.destroy_process_group()


def test_two_ranks_tree_to_hashes_yaml(oracle, tmp_path, snaphash_mode):
    """The whole pass as bench.py --gpus N times it (ABI 4 snaphash_shard_*): two ranks, each its LPT share of ONE on-disk
    tree through the HIP kernels (gpu_only) or as each rank's library plans its share (planned: a rank plans with ITS
    share of the node's cores), slabs gathered, hashes.yaml on every rank byte-identical to the oracle's."""
    import torch.multiprocessing as mp
    import trees
    rng = np.random.default_rng(21)
    sizes = [int(x) for x in rng.integers(0, 200000, size=300)] + [0, 1, 127, 128, 129, 3 << 20, (1 << 20) + 7, 1 << 20]
    build, tar = trees.make_synthetic_tree(str(tmp_path), sizes)
    os.makedirs(os.path.join(build, "DEBIAN"))
    open(os.path.join(build, "DEBIAN", "control"), "w").write("x")
    want = oracle.hashes_yaml(build, tar)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tree_worker, args=(r, 2, port, build, tar, q, snaphash_mode == "planned")) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[2] for r in res) == len(sizes)
    assert sum(r[3] + r[4] for r in res) == sum(sizes)
    if snaphash_mode == "gpu_only":
        assert all(r[4] == 0 for r in res)  # every byte through the kernels
    for rank, y, *_ in res:
        assert y == want, rank
