#!/usr/bin/env python3
"""bench.py -- the hashes.yaml SHA-512 pass on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A step is one pass of the hot path over one batch of synthetic input: every
file of ONE tree hashed by the HIP kernels from bytes already resident in HBM
(snaphash_sha512_device), plus -- for N > 1 -- the RCCL all-gather of the digest
vector.  The workload is what BASELINE.json's metric is quoted on: the
10 000 x 1 MiB tree plus its 1 MiB archive stand-in (10 001 streams), at
N = 1 (config 2) and LPT-sharded over N GPUs (config 4: the SAME one tree, so
`scaling` is "strong" and the N = 1 value of a scaling run equals the plain
N = 1 bench).  --scaling weak tiles N trees instead (labelled side experiment).

Prints ONE JSON line on rank 0:
  value         whole-job GiB/s with inputs resident in HBM when the timed region
                starts (value_kind says so; the PCIe- and disk-inclusive rates are
                in `end_to_end`, they are never `value`)
  roofline      the dominant kernel against the 8 TB/s HBM-read roofline
                (algorithmic bytes = file bytes hashed); kernel time from HIP
                events on the launch stream (library stats)
  end_to_end    every N: `buffers_sharded` -- each rank hashes ITS LPT shard of the one
                tree from host memory through snaphash_sha512_buffers (its own PCIe
                link, staging on the GPU's NUMA node), timed between barriers, MAX over
                ranks: the one rate of this path that shards.
                N = 1 also: on-disk tree -> hashes.yaml (snaphash_tree, compared byte
                for byte with the oracle's), `package` (a tree beside an archive of its
                own size, the library's DEFAULT configuration) and `build`: Build's data
                step fused (tar + GPU DEFLATE + archive digest + hashes.yaml) on a 1 GiB
                text tree
  cpu_baseline  the oracle (C restatement of the reference's serial loop) timed
                on this box's host cores over a bounded sample of the same tree
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# measured ceilings of this implementation (profiles/); refreshed per round
VALU_SATURATED_GBPS = float(os.environ.get("SNAPHASH_VALU_CEILING_GBPS", "1215"))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C3", "C5"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--kernel", default="auto", choices=["auto", "wide", "split", "pair", "quad"])
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="CPU work budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--e2e", default="auto", choices=["auto", "off", "buffers", "full"],
                    help="end_to_end legs at N = 1: auto = full for C2, buffers otherwise")
    ap.add_argument("--e2e-files", type=int, default=0, help="files of the on-disk tree leg (0 = the whole tree)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------
# cpu_baseline: the oracle on host cores (reported baseline, not the target)
# ------------------------------------------------------------------------------------------
def cpu_pool_leg(sizes, seconds):
    """The same C port on a pool of host threads (ctypes releases the GIL): what an embarrassingly
    parallel rewrite of the reference's loop would reach.  A 1-GPU box grants ~16 CPUs whatever
    os.cpu_count() says, so the pool is capped there; time-bounded."""
    from oracle import oracle
    from concurrent.futures import ThreadPoolExecutor
    n = len(sizes)
    cores = min(16, os.cpu_count() or 1)
    blobs = [oracle.fill_synthetic(int(sizes[k % n]), k % n) for k in range(cores)]
    deadline = time.perf_counter() + seconds

    def work(k):
        b = blobs[k]
        off = np.zeros(1, dtype=np.uint64)
        ln = np.array([len(b)], dtype=np.uint64)
        done = 0
        while time.perf_counter() < deadline:
            oracle.sha512_batch(b, off, ln)
            done += len(b)
        return done
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        tot = sum(ex.map(work, range(cores)))
    return {"value": round(tot / (time.perf_counter() - t0) / 2**30, 3), "cores": cores,
            "note": "same C port on a %d-thread pool (the reference itself is single-goroutine)" % cores}


def cpu_baseline_buffers(sizes, budget_s):
    """Files of the workload, in order, one at a time like the reference's filepath.Walk loop,
    until ~budget_s of CPU work (content generation not timed)."""
    from oracle import oracle
    n = len(sizes)
    done_bytes, files, t_hash, i = 0, 0, 0.0, 0
    while t_hash < budget_s and i < n:
        data = oracle.fill_synthetic(int(sizes[i]), i)
        off = np.zeros(1, dtype=np.uint64)
        ln = np.array([len(data)], dtype=np.uint64)
        t0 = time.perf_counter()
        oracle.sha512_batch(data, off, ln)
        t_hash += time.perf_counter() - t0
        done_bytes += len(data)
        files += 1
        i += 1
    return {"value": round(done_bytes / t_hash / 2**30, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
            "sample": "first %d files of the workload (%.1f MiB) through the oracle's Sha512sum loop, serial, "
                      "%.1f s of CPU work; content generation not timed" % (files, done_bytes / 2**20, t_hash),
            "host_cores_visible": os.cpu_count()}


# ------------------------------------------------------------------------------------------
# end_to_end legs (N = 1): the whole pass as a caller sees it
# ------------------------------------------------------------------------------------------
def e2e_buffers_sharded(ctx, host, offsets, lens, want_digests, total_bytes, world, fence, allmax, gather):
    """Every rank: ITS shard of the one tree, host memory in, digests out (pinned staging on the GPU's NUMA node,
    double-buffered H2D over the rank's own PCIe link, kernels).  Timed between barriers, MAX over ranks; the value is
    the whole tree's bytes over that time.  At N = 1 this is the plain host-buffers leg."""
    import ctypes
    from snappy_amd import _lib
    n = len(lens)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[host.ctypes.data + int(o) for o in offsets])
    clens = (ctypes.c_uint64 * max(n, 1))(*[int(x) for x in lens])
    out = ctypes.create_string_buffer(64 * max(n, 1))
    best = None
    for _ in range(3):
        fence()
        t0 = time.perf_counter()
        rc = _lib.lib().snaphash_sha512_buffers(ctx._h, ptrs, clens, n, out)
        mine = time.perf_counter() - t0
        fence()
        dt = allmax(time.perf_counter() - t0)
        if rc:
            raise SystemExit("snaphash_sha512_buffers failed: %d" % rc)
        st = ctx.stats()
        if best is None or dt < best[0]:
            best = (dt, mine, st)
    got = np.frombuffer(out.raw, dtype=np.uint8)[:64 * n].reshape(n, 64)
    if not np.array_equal(got, want_digests):
        raise SystemExit("PARITY FAILURE: host-buffer digests differ from the HBM-resident pass")
    dt, mine, st = best
    info = ctx.engine_info(0)
    per_rank = gather({"ms": round(mine * 1e3, 2), "h2d_ms": round(st["h2d_ms"], 2), "kernel_ms": round(st["kernel_ms"], 2),
                       "bytes": int(np.sum(lens)), "streams": n, "launches": int(st["launches"]),
                       "numa_node": info["numa_node"], "staging_node": info["staging_node"], "fill_threads": info["fill_threads"]})
    out = {"what": "host buffers -> snaphash_sha512_buffers -> digests on the host, every rank its LPT shard of the ONE tree "
                   "over its own PCIe link (pinned staging on the GPU's NUMA node, H2D, kernels); barrier to barrier, MAX over ranks",
           "n_gpus": world, "ms": round(dt * 1e3, 2), "GiBps": round(total_bytes / 2**30 / dt, 2),
           "h2d_ms": round(st["h2d_ms"], 2), "kernel_ms": round(st["kernel_ms"], 2), "launches": int(st["launches"]),
           "per_rank": per_rank, "every_byte_on_the_gpu": True,
           "parity": "every rank's digest vector identical to its HBM-resident pass", "best_of": 3}
    return out


def e2e_package(total_mib=512):
    """A package as `snappy build` meets it: a tree and, beside it, an archive of the tree's own size (the stand-in
    for its data.tar.gz, snappy/build.go:222) through snaphash_tree in the library's DEFAULT configuration -- the
    archive is ONE stream, so the default hands it to a host thread while the GPU takes the tree.  hashes.yaml is
    compared with the oracle's."""
    from oracle import oracle
    from snappy_amd import Context, synthetic
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    if base and shutil.disk_usage(base).free < (total_mib << 20) * 2 + (1 << 30):
        total_mib = max(32, int((shutil.disk_usage(base).free - (1 << 30)) // (2 << 20)))
    tmp = tempfile.mkdtemp(prefix="snaphash_pkg_", dir=base)
    try:
        build = os.path.join(tmp, "build")
        rng = np.random.default_rng(9)
        blob = rng.integers(0, 256, size=(2 << 20) + 64, dtype=np.uint8).tobytes()
        for i in range(total_mib):
            p = os.path.join(build, synthetic.file_name(i))
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "wb") as f:
                f.write(blob[i % 4096:(i % 4096) + (1 << 20)])
        tar = os.path.join(tmp, "data.tar.gz")
        with open(tar, "wb") as f:
            for i in range(total_mib):
                f.write(blob[(7 * i) % 4096:((7 * i) % 4096) + (1 << 20)])
        with Context(flags=0) as c:  # the defaults a cgo caller gets from snaphash_init(NULL)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                y = c.tree(build, tar)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, c.stats(), c.stats_ex())
        dt, st, ex = best
        t0 = time.perf_counter()
        want = oracle.hashes_yaml(build, tar)
        dt_c = time.perf_counter() - t0
        if want != y:
            raise SystemExit("PARITY FAILURE: package leg: hashes.yaml differs from the oracle's")
        total = 2 * (total_mib << 20)
        return {"what": "%d x 1 MiB tree (tmpfs) + a %d MiB archive beside it -> snaphash_tree, DEFAULT configuration "
                        "(snaphash_init(NULL)): the archive, one stream, on a host thread; the tree on the GPU" % (total_mib, total_mib),
                "bytes": total, "ms": round(dt * 1e3, 1), "GiBps": round(total / 2**30 / dt, 2),
                "gpu_bytes": int(ex["gpu_bytes"]), "host_bytes": int(ex["host_bytes"]), "host_streams": int(ex["host_streams"]),
                "host_ms": round(ex["host_ms"], 1), "kernel_ms": round(st["kernel_ms"], 1),
                "cpu_port_serial_ms": round(dt_c * 1e3, 1),
                "parity": "hashes.yaml byte-identical to the oracle's", "best_of": 3}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def e2e_tree(ctx, host, offsets, lens, nfiles, cpu_seconds):
    """on-disk tree -> hashes.yaml: walk, parallel pread, staging, H2D, kernels, YAML; the oracle's
    serial CPU pass over the same tree is both the checker and the reference-shaped timing."""
    from oracle import oracle
    from snappy_amd import synthetic
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    n = min(nfiles, len(lens) - 1) if nfiles else len(lens) - 1
    need = int(np.sum(lens[:n])) + int(lens[-1])
    if base:
        free = shutil.disk_usage(base).free
        if free < need + (2 << 30):
            n = max(100, int(n * (free - (2 << 30)) / max(need, 1)))
    tmp = tempfile.mkdtemp(prefix="snaphash_bench_", dir=base)
    try:
        build = os.path.join(tmp, "build")
        os.makedirs(build)
        made = set()
        for i in range(n):
            p = os.path.join(build, synthetic.file_name(i))
            d = os.path.dirname(p)
            if d not in made:
                os.makedirs(d, exist_ok=True)
                made.add(d)
            host[int(offsets[i]):int(offsets[i]) + int(lens[i])].tofile(p)
        tar = os.path.join(tmp, "data.tar.gz")
        host[int(offsets[-1]):int(offsets[-1]) + int(lens[-1])].tofile(tar)
        total = int(np.sum(lens[:n])) + int(lens[-1])
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            y_gpu = ctx.tree(build, tar)
            dt = time.perf_counter() - t0
            st = ctx.stats()
            if best is None or dt < best[0]:
                best = (dt, st)
        t0 = time.perf_counter()
        res = ctx.verify(build, y_gpu, tar)
        dt_v = time.perf_counter() - t0
        if res is not None:
            raise SystemExit("Verify reports a mismatch on the tree just hashed: %r" % (res,))
        out = {"what": "on-disk tree (tmpfs) -> snaphash_tree -> hashes.yaml (walk, pread, staging, H2D, kernels, YAML)",
               "files": n + 1, "bytes": total, "ms": round(best[0] * 1e3, 2), "GiBps": round(total / 2**30 / best[0], 2),
               "h2d_ms": round(best[1]["h2d_ms"], 2), "kernel_ms": round(best[1]["kernel_ms"], 2),
               "verify_ms": round(dt_v * 1e3, 2), "yaml_bytes": len(y_gpu), "best_of": 3}
        cpu = None
        if cpu_seconds > 0:
            t0 = time.perf_counter()
            y_cpu = oracle.hashes_yaml(build, tar)
            dt_c = time.perf_counter() - t0
            if y_cpu != y_gpu:
                raise SystemExit("PARITY FAILURE: hashes.yaml differs from the oracle's")
            out["parity"] = "hashes.yaml byte-identical to the oracle's (%d records)" % y_gpu.count(b"- name: ")
            cpu = {"value": round(total / 2**30 / dt_c, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
                   "sample": "the oracle's whole writeHashes pass (walk, 32 KiB reads, SHA-512, YAML) over the same "
                             "on-disk tree of %d files / %.0f MiB: %.1f s on one core, as the reference's single "
                             "goroutine would run it" % (n + 1, total / 2**20, dt_c),
                   "host_cores_visible": os.cpu_count()}
        else:
            out["parity"] = "not checked (--cpu-seconds 0)"
        return out, cpu
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def deflate_roofline(zs):
    """roofline object of deflate_chunks_kernel (+ the concatenation kernel: deflate_ms is both): algorithmic bytes =
    tar bytes read + gz bytes written, over the kernels' HIP-event time; counter traffic from the profile of the same
    workload when one is committed (profiles/traffic_deflate_chunks_kernel.json)."""
    alg = zs["tar_bytes"] + zs["gz_bytes"]
    ach = alg / (zs["deflate_ms"] * 1e-3) / 1e9
    traffic, src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_deflate_chunks_kernel.json")
    if os.path.exists(tpath):  # counters of the same workload (Zipf-word text, 1 MiB files), per input byte
        tj = json.load(open(tpath))
        traffic = {"hbm_read_bytes": int(tj["read_per_input_byte_upper"] * zs["tar_bytes"]),
                   "hbm_write_bytes": int(tj["write_per_input_byte"] * zs["tar_bytes"]),
                   "read_per_input_byte": round(tj["read_per_input_byte_upper"], 2), "write_per_input_byte": round(tj["write_per_input_byte"], 2),
                   "note": "FETCH_SIZE x 2 (upper reading: the correction is calibrated for 16 B/lane streaming only), WRITE_SIZE as read"}
        src = os.path.relpath(tpath, ROOT)
    return {"bound": "hbm", "kernel": "deflate_chunks_kernel (+ deflate_compact_kernel)", "achieved": round(ach, 2),
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5), "algorithmic_bytes": int(alg),
            "traffic": traffic, "traffic_source": src, "GBps_of_input": round(zs["tar_bytes"] / (zs["deflate_ms"] * 1e-3) / 1e9, 2),
            "note": "algorithmic bytes = input read + output written; bound by instruction issue and memory latency of the chain walk (DESIGN.md sec. 9)"}


def e2e_build(ctx, total_mib=1024):
    """Rows f2 + f3: `Build`'s data step in one pass -- data.tar.gz (GPU DEFLATE) + archive digest + per-file SHA-512
    + hashes.yaml, every file read once -- on a compressible tree (Zipf-word text, 1 MiB files); the archive is read
    back with tarfile and the yaml compared with the oracle's over the tree and the archive just written."""
    import tarfile
    from oracle import oracle
    from snappy_amd import synthetic
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    if base and shutil.disk_usage(base).free < (total_mib << 20) * 2 + (1 << 30):
        total_mib = max(64, int((shutil.disk_usage(base).free - (1 << 30)) // (2 << 20)))  # the tree and its archive must fit
    tmp = tempfile.mkdtemp(prefix="snaphash_build_", dir=base)
    try:
        build = os.path.join(tmp, "build")
        os.makedirs(os.path.join(build, "DEBIAN"))
        rng = np.random.default_rng(5)
        words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
        block = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=(4 << 20) // 5 + 16) % 2000)[:4 << 20]
        for i in range(total_mib):
            p = os.path.join(build, synthetic.file_name(i))
            os.makedirs(os.path.dirname(p), exist_ok=True)
            off = int(rng.integers(0, len(block) - 1))
            with open(p, "wb") as f:
                f.write((block[off:] + block)[:1 << 20])
        out = os.path.join(tmp, "data.tar.gz")
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            y, dig = ctx.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, ctx.stats(), ctx.targz_stats())
        dt, st, zs = best
        if hashlib.sha512(open(out, "rb").read()).digest() != dig or oracle.hashes_yaml(build, out) != y:
            raise SystemExit("PARITY FAILURE: the fused build pass disagrees with hashlib / the oracle")
        tf = tarfile.open(out, "r:gz")
        for k, m in enumerate(tf):
            if k >= 24:
                break
            if m.isreg() and tf.extractfile(m).read() != open(os.path.join(build, m.name[2:]), "rb").read():
                raise SystemExit("PARITY FAILURE: archive member %s differs from the file" % m.name)
        return {"what": "on-disk tree (tmpfs, %d x 1 MiB Zipf-word text) -> snaphash_tar_create: tar + GPU DEFLATE + archive "
                        "SHA-512 + per-file SHA-512 + hashes.yaml, one read of every file" % total_mib,
                "tar_bytes": int(zs["tar_bytes"]), "gz_bytes": int(zs["gz_bytes"]), "ratio": round(zs["gz_bytes"] / zs["tar_bytes"], 4),
                "ms": round(dt * 1e3, 1), "GiBps_of_tree": round(zs["tar_bytes"] / 2**30 / dt, 2),
                "deflate_kernel_ms": round(zs["deflate_ms"], 1), "sha512_kernel_ms": round(st["kernel_ms"], 1),
                "deflate_kernel": deflate_roofline(zs),
                "bound": "the DEFLATE kernel, with the serial SHA-512 of the archive on one host core close behind (DESIGN.md sec. 9)",
                "parity": "archive inflates to the tree (tarfile); archive digest = hashlib; hashes.yaml byte-identical to the oracle's",
                "best_of": 3}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = parse_args()
    # stdout carries ONE JSON line and nothing else: whatever the libraries below print there (RCCL's version banner
    # at the first collective, for one) goes to stderr
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from snappy_amd import Context, _lib, synthetic
    from snappy_amd.sharded import ShardPlan, gather_digests

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    # SNAPHASH_BENCH_FORCE_DIST=1: run the RCCL path even with one rank (rehearsal on a 1-GPU box)
    force_dist = os.environ.get("SNAPHASH_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    kern = {"auto": _lib.KERNEL_AUTO, "wide": _lib.KERNEL_WIDE, "split": _lib.KERNEL_SPLIT, "pair": _lib.KERNEL_PAIR,
            "quad": getattr(_lib, "KERNEL_QUAD", 4)}[args.kernel]
    stream = torch.cuda.current_stream().cuda_stream
    ctx = Context(device=local_rank, kernel=kern, stream=stream)
    tree = synthetic.config_sizes(args.workload)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_job(sizes, steps, warmup):
        """One file list LPT-sharded over the ranks; returns timing + the gathered digest matrix."""
        findex = np.arange(len(sizes), dtype=np.uint64)
        plan = ShardPlan(sizes, world)
        mine = plan.members(rank)
        my_lens = np.ascontiguousarray(sizes[mine])
        my_off, my_total = synthetic.pack_offsets(my_lens)
        data = torch.empty(max(my_total, 16), dtype=torch.uint8, device="cuda")
        ctx.fill_synthetic_device(data.data_ptr(), my_off, my_lens, np.ascontiguousarray(findex[mine]))
        local = torch.zeros((max(plan.kmax, 1), 64), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        kernel_ms = []
        full = None

        def step(record):
            ctx.sha512_device(data.data_ptr(), my_off, my_lens, local.data_ptr())
            out = gather_digests(local, plan, force_collective=force_dist)  # RCCL all-gather of the slabs (no-op at N = 1)
            ctx.sync()
            if record:
                kernel_ms.append(ctx.stats()["kernel_ms"])
            return out
        for _ in range(warmup):
            step(False)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            full = step(True)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return {"elapsed": elapsed, "kernel_ms": float(np.mean(kernel_ms)), "my_bytes": int(my_lens.sum()),
                "my_streams": len(my_lens), "digests": full.cpu().numpy(), "stats": ctx.stats(),
                "data": data, "my_off": my_off, "my_lens": my_lens, "total_bytes": int(sizes.sum())}

    ntrees = world if args.scaling == "weak" else 1
    sizes = np.tile(tree, ntrees)
    job = run_job(sizes, args.steps, args.warmup)

    # the other reading of "scaling" beside the headline, on record (N > 1 only)
    side_leg = None
    if use_dist:
        other = "weak" if args.scaling == "strong" else "strong"
        osizes = np.tile(tree, world if other == "weak" else 1)
        nrep = max(3, min(10, args.steps))
        j2 = run_job(osizes, nrep, 2)
        side_leg = {"scaling": other,
                    "workload": "%d %s tree(s) (%d files) LPT-sharded over %d GPU(s) + RCCL digest all-gather" %
                                (world if other == "weak" else 1, args.workload, len(osizes), world),
                    "value": round(j2["total_bytes"] / 2**30 / (j2["elapsed"] / nrep), 3), "unit": "GiB/s",
                    "ms_per_step": round(j2["elapsed"] / nrep * 1e3, 4), "steps": nrep}
        del j2

    # ---- parity spot check of the timed path, outside the timed region ------------------
    digests = job["digests"]
    parity = None
    if rank == 0:
        rng = np.random.default_rng(1)
        sample = sorted(set([0, len(sizes) - 1] + [int(x) for x in rng.integers(0, len(sizes), size=14)]))
        sample = [i for i in sample if sizes[i] <= (64 << 20)] or [int(np.argmin(sizes))]
        for i in sample:  # independent check: numpy generator + hashlib (OpenSSL), not the timed path
            want = hashlib.sha512(synthetic.file_bytes(int(sizes[i]), int(i))).digest()
            if digests[i].tobytes() != want:
                raise SystemExit("PARITY FAILURE: file %d digest differs from hashlib.sha512" % i)
        parity = {"checked_files": len(sample), "result": "bit-exact vs hashlib.sha512",
                  "sha512_of_digest_vector": hashlib.sha512(digests.tobytes()).hexdigest()[:32]}

    # ---- end-to-end legs ------------------------------------------------------------------
    # every N: each rank hashes its shard from host memory over its own PCIe link (the rate that shards);
    # N = 1 also: on-disk tree, package (default configuration), fused build; and the CPU baseline.
    end_to_end, cpu = None, None
    mode = args.e2e
    if mode == "auto":
        mode = "full" if args.workload == "C2" else ("buffers" if job["total_bytes"] <= (16 << 30) else "off")
    if force_dist and world == 1 and mode == "full":
        mode = "buffers"
    if mode != "off":
        # The source buffers belong on the socket this rank's GPU hangs off (first touch by a thread that runs there), as the
        # engine's staging memory and fill threads are (snaphash_get_engine_info): eight ranks' copies then stay on
        # their own memory controllers instead of crossing the socket link.
        numa_note = "not bound"
        try:
            probe = Context(device=local_rank, flags=_lib.FLAG_GPU_ONLY)
            node = probe.engine_info(0)["numa_node"]
            probe.close()
            if node >= 0:
                cpus = []
                for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
                    lo, _, hi = part.partition("-")
                    cpus += list(range(int(lo), int(hi or lo) + 1))
                allowed = sorted(set(cpus) & os.sched_getaffinity(0))
                if allowed:
                    os.sched_setaffinity(0, allowed)
                    numa_note = "rank bound to NUMA node %d (%d CPUs) before its source buffers were allocated" % (node, len(allowed))
        except Exception as e:  # noqa: BLE001  (placement is an optimisation: never a reason to lose the line)
            numa_note = "not bound: %r" % (e,)
        host = job["data"].cpu().numpy()  # the same bytes the resident pass hashed, now in this rank's host memory
        del job["data"]
        torch.cuda.empty_cache()
        ectx = Context(device=local_rank, kernel=kern, flags=_lib.FLAG_GPU_ONLY)  # own stream and staging engine
        end_to_end = {}

        def allmax(x):
            if not use_dist:
                return x
            t = torch.tensor([x], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        def gather(obj):
            if not use_dist:
                return [obj]
            objs = [None] * world
            dist.all_gather_object(objs, obj)
            return objs
        mine = ShardPlan(sizes, world).members(rank)
        local_want = digests[mine] if len(mine) else digests[:0]
        r = e2e_buffers_sharded(ectx, host, job["my_off"], job["my_lens"], local_want, job["total_bytes"], world, fence, allmax, gather)
        r["source_placement"] = numa_note
        end_to_end["buffers_sharded"] = r
        if world == 1:
            end_to_end["buffers"] = {k: v for k, v in r.items() if k != "per_rank"}  # the name round 2 reported this leg under

        def leg(name, fn):
            # an auxiliary leg that cannot run here (no room in /dev/shm, ...) is recorded, it never costs the headline
            # line; a PARITY FAILURE is a SystemExit and still ends the run
            try:
                return fn()
            except Exception as e:  # noqa: BLE001
                end_to_end[name] = {"error": repr(e)[:300]}
                return None
        if rank == 0 and world == 1 and mode == "full":
            r = leg("tree", lambda: e2e_tree(ectx, host, job["my_off"], job["my_lens"], args.e2e_files, args.cpu_seconds))
            if r is not None:
                end_to_end["tree"], cpu = r
            del host
            r = leg("package", e2e_package)
            if r is not None:
                end_to_end["package"] = r
            r = leg("build", lambda: e2e_build(ectx))
            if r is not None:
                end_to_end["build"] = r
        ectx.close()
    if rank == 0 and world == 1 and not force_dist and args.cpu_seconds > 0:
        if cpu is None:
            cpu = cpu_baseline_buffers(tree, args.cpu_seconds)
        cpu["all_cores"] = cpu_pool_leg(tree, 3.0)

    if rank == 0:
        st = job["stats"]
        ms_step = job["elapsed"] / args.steps * 1e3
        value = job["total_bytes"] / 2**30 / (job["elapsed"] / args.steps)
        k_ms = job["kernel_ms"]
        achieved = job["my_bytes"] / (k_ms * 1e-3) / 1e9
        kname = _lib.KERNEL_NAMES.get(st["kernel_used"], "sha512_wide_kernel")
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % kname.replace("<", "_").replace(">", ""))
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            # PMC traffic is a per-launch figure of ONE workload: only quote it for that workload
            if tj.get("workload", "C2") == args.workload and world == 1 and tj.get("bytes_per_launch", job["my_bytes"]) == job["my_bytes"]:
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), os.path.relpath(tpath, ROOT)
        wl = "%s: %s%d files%s, HBM-resident, LPT-sharded over %d GPU(s)%s" % (
            args.workload, "ONE tree of " if ntrees == 1 else "%d trees, " % ntrees, len(sizes),
            " (10 000 x 1 MiB + the 1 MiB archive stand-in)" if args.workload == "C2" and ntrees == 1 else "",
            world, ", RCCL all-gather of the digest vector" if world > 1 else "")
        line = {
            "metric": "GiB/s hashed (whole node), bit-exact hashes.yaml, 10k x 1 MiB tree" if args.workload == "C2" and ntrees == 1
                      else "GiB/s hashed (whole node), bit-exact hashes.yaml, workload %s x%d" % (args.workload, ntrees),
            "value": round(value, 3), "unit": "GiB/s",
            "value_kind": "hbm_resident: file bytes already in HBM when the timed region starts; kernels + digest gather "
                          "only.  This job is stream-count-bound and flat in N by construction (DESIGN.md sec. 5); the rate "
                          "that scales with N is end_to_end.buffers_sharded (host memory -> N PCIe links -> digests)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": wl, "files": int(len(sizes)), "bytes": job["total_bytes"], "kernel": kname,
                       "launches_per_step": int(st["launches"]),
                       "sha512_blocks_per_step": int(st["blocks"]) if world == 1 else None},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "kernel_ms": round(k_ms, 4), "bytes_per_launch": job["my_bytes"],
                         "frac_of_valu_ceiling": round(achieved / VALU_SATURATED_GBPS, 4),
                         "note": "algorithmic bytes = file bytes hashed by rank 0's launch; SHA-512 is integer-VALU "
                                 "and stream-count bound, not HBM bound (DESIGN.md sec. 4)"},
            "parity": parity,
            "ceilings": {"hbm_GBps": HBM_PEAK_GBPS, "valu_saturated_GBps_measured": VALU_SATURATED_GBPS,
                         "per_stream_MBps_here": round(achieved * 1e3 / max(job["my_streams"], 1), 2),
                         "source": "profiles/ (regime sweep, saturated-launch PMC)"},
        }
        if side_leg is not None:
            line["other_scaling_leg"] = side_leg
        if end_to_end is not None:
            line["end_to_end"] = end_to_end
        if cpu is not None:
            line["cpu_baseline"] = cpu
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # ---- N > 1: the same tree through ONE process and the C ABI's device list (what the Go caller reaches over
        # cgo): in-library LPT shards, one NUMA-placed staging engine per GPU, single-process RCCL gather.  Run when
        # the ranks have let go of their GPUs, in a child process with a time limit: the line never depends on it. ----
        if world > 1 and os.environ.get("SNAPHASH_BENCH_NO_INLIB") != "1" and args.workload in ("C1", "C2", "C5"):
            import subprocess
            torch.cuda.empty_cache()
            try:
                env = dict(os.environ)
                for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK",
                          "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE"):
                    env.pop(k, None)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "inlib_multigpu.py"), args.workload, "-1"],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, env=env)
                lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
                inlib = json.loads(lines[-1]) if lines else {"error": "rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-400:])}
            except Exception as e:  # a time-out or a missing RCCL must not cost the headline
                inlib = {"error": repr(e)[:400]}
            if "sha512_of_digest_vector" in inlib:
                inlib["same_digest_vector_as_the_timed_path"] = inlib["sha512_of_digest_vector"] == parity["sha512_of_digest_vector"]
            line["in_library_multi_gpu"] = inlib
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()


if __name__ == "__main__":
    main()
