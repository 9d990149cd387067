#!/usr/bin/env python3
"""bench.py -- the hashes.yaml SHA-512 pass on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU over RCCL.  Under an outer launcher -- torch.distributed.run sets WORLD_SIZE / RANK /
     LOCAL_RANK -- this process IS a rank; without one the plain command starts its N ranks itself, as child
     processes, before anything in the parent has touched a GPU: self_launch)

The metric is "GiB/s hashed (whole node), bit-exact hashes.yaml" on the 10 000 x 1 MiB tree (10 001 streams with its
1 MiB archive stand-in).  SURVEY sec. 8(d) names two timings and says which one that is:

  (ii) END TO END -- `value`.  A step is one writeHashes pass (snappy/build.go:216-270) over ONE on-disk tree (tmpfs):
       walk, pread into pinned staging, H2D over PCIe, the HIP kernels, hashes.yaml.  Every byte is hashed by the
       kernels (SNAPHASH_FLAG_GPU_ONLY).  N = 1: snaphash_tree.  N > 1: the same tree, every rank its LPT share
       (snaphash_shard_plan / _hash), ONE RCCL all-gather of the digest slabs, rank 0 writes hashes.yaml
       (snaphash_shard_emit); `scaling` is "strong", barrier + synchronize on both sides, MAX over ranks.
  (i)  KERNEL-RESIDENT -- `roofline` (and `hbm_resident`): the same streams already in HBM, snaphash_sha512_device; the
       dominant kernel's algorithmic bytes over its HIP-event time against the 8 TB/s HBM-read roofline.  This rate is
       stream-count-bound and flat in N by construction; it is reported, it is not `value`.

Prints ONE JSON line on rank 0.  Beside the two above:
  parity        hashes.yaml of the timed path byte-identical to the oracle's (every N; at N > 1 also to the single-GPU
                pass), and a hashlib sample
  config.ranks  world size, backend, launcher and every rank's GPU by PCI bus id: N distinct GPUs, or the line says so
  end_to_end    buffers_sharded (host buffers -> digests, every N); N = 1 also: tree_default (the library's DEFAULT
                configuration: the planner may give host cores a share), package, build (rows f2 + f3, with the
                reference-shaped and the all-cores zlib baselines), breakeven (small calls against the reference's loop)
  configs       BASELINE configs 1, 3 and 5 on this GPU: HBM-resident GPU-only pass (roofline per config), the DEFAULT
                configuration from host memory, the CPU port beside it, sampled parity vs hashlib
  cpu_baseline  the oracle (C restatement of the reference's serial loop) over the same on-disk tree on one core (every N)
"""
import argparse
import ctypes
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
GiB = float(1 << 30)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C5"], help="the tree of the timed pass (BASELINE's metric: C2)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "wide", "split", "pair", "quad"])
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="0 = skip the cpu_baseline leg (the oracle's serial pass over the whole tree, ~21 s for C2)")
    ap.add_argument("--legs", default="auto", choices=["auto", "off", "full"],
                    help="auto = every leg for the default C2 run at N = 1, buffers_sharded only otherwise")
    ap.add_argument("--resident-steps", type=int, default=12, help="launches of the HBM-resident (roofline) pass")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------
def shm_dir():
    return "/dev/shm" if os.path.isdir("/dev/shm") else None


def bind_to_gpu_node(local_rank):
    """The rank's source buffers and the tmpfs pages it writes belong on the socket its GPU hangs off (first touch by a
    thread that runs there), as the engine's staging memory and fill threads are (snaphash_get_engine_info)."""
    from snappy_amd import Context, _lib
    try:
        probe = Context(device=local_rank, flags=_lib.FLAG_GPU_ONLY)
        node = probe.engine_info(0)["numa_node"]
        probe.close()
        if node < 0:
            return "not bound (single NUMA node or unknown)"
        cpus = []
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus += list(range(int(lo), int(hi or lo) + 1))
        allowed = sorted(set(cpus) & os.sched_getaffinity(0))
        if not allowed:
            return "not bound (node %d has none of this process's CPUs)" % node
        os.sched_setaffinity(0, allowed)
        return "rank bound to NUMA node %d (%d CPUs) before it wrote its files / allocated its buffers" % (node, len(allowed))
    except Exception as e:  # noqa: BLE001  (placement is an optimisation: never a reason to lose the line)
        return "not bound: %r" % (e,)


def file_index_of(path, n_files):
    """d0012/f001234.bin -> 1234; the archive stand-in (data.tar.gz) -> n_files."""
    base = os.path.basename(path)
    return int(base[1:7]) if base.startswith("f") and base.endswith(".bin") else n_files


def hashlib_check(paths, digests_hex, what):
    for p, want in zip(paths, digests_hex):
        h = hashlib.sha512()
        with open(p, "rb") as f:
            for blk in iter(lambda: f.read(1 << 22), b""):
                h.update(blk)
        if h.hexdigest() != want:
            raise SystemExit("PARITY FAILURE (%s): %s differs from hashlib.sha512" % (what, p))


def yaml_digest_of(yaml_bytes, name):
    """sha512 text of the record `name` in a hashes.yaml written by the library (plain names only)."""
    key = b"- name: " + name.encode() + b"\n"
    at = yaml_bytes.index(key)
    line = yaml_bytes.index(b"  sha512: ", at)
    return yaml_bytes[line + 10:line + 138].decode()


# ------------------------------------------------------------------------------------------
# cpu_baseline helpers: the oracle on host cores (reported baseline, not the target)
# ------------------------------------------------------------------------------------------
def cpu_pool_leg(sizes, seconds):
    """The same C port on a pool of host threads (ctypes releases the GIL): what an embarrassingly parallel rewrite of the
    reference's loop would reach on the cores this job may use; time-bounded."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from snappy_amd import _lib
    n = len(sizes)
    cores = max(1, min(int(_lib.lib().snaphash_usable_cpus()), 64))
    blobs = [oracle.fill_synthetic(int(min(sizes[k % n], 8 << 20)), k % n) for k in range(cores)]
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
    return {"value": round(tot / (time.perf_counter() - t0) / GiB, 3), "cores": cores,
            "note": "same C port on a %d-thread pool = the cores this job may keep busy (cgroup quota; the reference itself is single-goroutine)" % cores}


def cpu_port_sample(host, offsets, lens, budget_s, order=None):
    """The oracle's Sha512sum loop, serial, over files of the workload until ~budget_s of CPU work."""
    from oracle import oracle
    n = len(lens)
    done_bytes, files, t_hash = 0, 0, 0.0
    for i in (order if order is not None else range(n)):
        if t_hash >= budget_s:
            break
        off = np.array([int(offsets[i])], dtype=np.uint64)
        ln = np.array([int(lens[i])], dtype=np.uint64)
        t0 = time.perf_counter()
        oracle.sha512_batch(host, off, ln)
        t_hash += time.perf_counter() - t0
        done_bytes += int(lens[i])
        files += 1
    return {"value": round(done_bytes / max(t_hash, 1e-9) / GiB, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
            "sample": "%d file(s) of the workload (%.1f MiB) through the oracle's Sha512sum loop, serial, %.2f s of CPU work" %
                      (files, done_bytes / 2**20, t_hash)}


# ------------------------------------------------------------------------------------------
# the HBM-resident pass: the roofline object of a list of streams
# ------------------------------------------------------------------------------------------
def resident_pass(ctx, data_ptr, offsets, lens, slab_ptr, launches, warmup):
    """snaphash_sha512_device `launches` times after `warmup`; per-launch kernel time from HIP events on the launch stream
    (library stats), wall time per launch beside it."""
    for _ in range(warmup):
        ctx.sha512_device(data_ptr, offsets, lens, slab_ptr)
        ctx.sync()
    k_ms, t0 = [], time.perf_counter()
    st = None
    for _ in range(launches):
        ctx.sha512_device(data_ptr, offsets, lens, slab_ptr)
        ctx.sync()
        st = ctx.stats()
        k_ms.append(st["kernel_ms"])
    wall = (time.perf_counter() - t0) / max(launches, 1)
    return float(np.mean(k_ms)), wall, st


def roofline_of(kernel_ms, nbytes, st, workload, world, note=None, streams=None):
    from snappy_amd import _lib
    achieved = nbytes / (kernel_ms * 1e-3) / 1e9
    if world > 1:
        scope = ("PER GPU: rank 0's shard (%s streams, %d bytes) resident in ITS HBM against ONE GPU's 8 TB/s; stream-count-bound -- "
                 "a shard of 1/%d of the streams advances no faster per stream, so achieved and frac fall as 1/N by construction "
                 "while the launch takes as long as at N = 1 (DESIGN.md sec. 5)" % (streams if streams is not None else "?", int(nbytes), world))
    else:
        scope = "per GPU (= the whole job at N = 1): every stream of the workload resident in this GPU's HBM"
    kname = _lib.KERNEL_NAMES.get(st["kernel_used"], "sha512_wide_kernel")
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % kname.replace("<", "_").replace(">", ""))
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        # PMC traffic is a per-launch figure of ONE workload: only quote it for that workload
        if tj.get("workload", "C2") == workload and world == 1 and tj.get("bytes_per_launch", nbytes) == nbytes:
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), os.path.relpath(tpath, ROOT)
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "scope": scope, "n_gpus": world, "kernel": kname, "kernel_ms": round(kernel_ms, 4), "bytes_per_launch": int(nbytes),
            "launches_per_pass": int(st["launches"]), "sha512_blocks_per_pass": int(st["blocks"]),
            "frac_of_valu_ceiling": round(achieved / VALU_SATURATED_GBPS, 4),
            "note": note or "algorithmic bytes = file bytes hashed by rank 0's launch, inputs resident in HBM; SHA-512 is "
                            "integer-VALU and stream-count bound, not HBM bound (DESIGN.md sec. 4)"}


# ------------------------------------------------------------------------------------------
# the planner's prediction beside what the call took (snaphash_stats_ex, ABI 5)
# ------------------------------------------------------------------------------------------
def model_vs_actual(ex, tolerance=0.25):
    """the plan's modelled makespans next to the measured ones.  `off` names every side whose
    prediction missed by more than `tolerance` -- on another box (PCIe generation, CPU, quota) that is the sign that the
    model's constants do not hold there; the calibration (planner.h PlanCalib) then moves them."""
    rows = {}
    off = []
    for side, planned, actual in (("gpu", ex.get("planned_gpu_ms", 0.0), ex.get("gpu_ms", 0.0)),
                                  ("host", ex.get("planned_host_ms", 0.0), ex.get("host_ms", 0.0))):
        if planned <= 0 and actual <= 0:
            continue
        rows[side] = {"planned_ms": round(planned, 3), "actual_ms": round(actual, 3)}
        if planned > 0 and actual > 0:
            rows[side]["actual_over_planned"] = round(actual / planned, 3)
            # (sub-millisecond parts are start-up noise either way)
            if max(planned, actual) >= 1.0 and not (1 - tolerance) <= actual / planned <= 1 / (1 - tolerance):
                off.append(side)
    whole_p = max(ex.get("planned_gpu_ms", 0.0), ex.get("planned_host_ms", 0.0))
    rows["call"] = {"planned_ms": round(whole_p, 3), "hash_ms": round(ex.get("hash_ms", 0.0), 3), "plan_ms": round(ex.get("plan_ms", 0.0), 3),
                    "threads_planned": int(ex.get("planned_threads", 0)), "threads_run": int(ex.get("host_threads_run", 0))}
    if whole_p > 0 and ex.get("hash_ms", 0.0) > 0:
        rows["call"]["actual_over_planned"] = round(ex["hash_ms"] / whole_p, 3)
        if max(whole_p, ex["hash_ms"]) >= 1.0 and not (1 - tolerance) <= ex["hash_ms"] / whole_p <= 1 / (1 - tolerance):
            off.append("call")
    rows["off_by_more_than_%d_pct" % int(tolerance * 100)] = off
    return rows


# ------------------------------------------------------------------------------------------
# end_to_end legs
# ------------------------------------------------------------------------------------------
def e2e_buffers_sharded(ctx, host, offsets, lens, want_digests, total_bytes, world, fence, allmax, gather):
    """Every rank: ITS shard of the one tree, host memory in, digests out (pinned staging on the GPU's NUMA node,
    H2D over the rank's own PCIe link, kernels).  Timed between barriers, MAX over ranks; the value is the whole tree's
    bytes over that time."""
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
    return {"what": "host buffers -> snaphash_sha512_buffers -> digests on the host, every rank its LPT shard of the ONE tree "
                    "over its own PCIe link (pinned staging on the GPU's NUMA node, H2D, kernels); barrier to barrier, MAX over ranks",
            "n_gpus": world, "ms": round(dt * 1e3, 2), "GiBps": round(total_bytes / GiB / dt, 2),
            "h2d_ms": round(st["h2d_ms"], 2), "kernel_ms": round(st["kernel_ms"], 2), "launches": int(st["launches"]),
            "per_rank": per_rank, "every_byte_on_the_gpu": True,
            "parity": "every rank's digest vector identical to its HBM-resident pass", "best_of": 3}


def e2e_tree_default(build, tar, total, want_yaml, device):
    """The same tree through snaphash_tree in the library's DEFAULT configuration (what snaphash_init(NULL) gives a cgo
    caller): the planner may hand the cores the staging fill leaves free a share of the streams."""
    from snappy_amd import Context
    with Context(device=device, flags=0) as c:
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            y = c.tree(build, tar)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, c.stats(), c.stats_ex())
        if want_yaml is not None and y != want_yaml:
            raise SystemExit("PARITY FAILURE: default-configuration hashes.yaml differs from the GPU-only pass")
        plan_model = dict(c.plan_model(True), calibration=c.calib())  # what this ctx plans file sources with, after these passes
    dt, st, ex = best
    return {"what": "the same on-disk tree -> snaphash_tree, DEFAULT configuration (snaphash_init(NULL)): planned, the host "
                    "cores beside the staging fill take a share (the library's own SHA-512, hostsha.cpp)",
            "ms": round(dt * 1e3, 2), "GiBps": round(total / GiB / dt, 2), "gpu_bytes": int(ex["gpu_bytes"]),
            "host_bytes": int(ex["host_bytes"]), "host_streams": int(ex["host_streams"]), "host_ms": round(ex["host_ms"], 1),
            "h2d_ms": round(st["h2d_ms"], 2), "kernel_ms": round(st["kernel_ms"], 2),
            "model_vs_actual": model_vs_actual(ex), "plan_model": plan_model,
            "parity": "hashes.yaml byte-identical to the GPU-only pass (and so to the oracle's)", "best_of": 5}


def e2e_package(total_mib=512):
    """A package as `snappy build` meets it: a tree and, beside it, an archive of the tree's own size (the stand-in
    for its data.tar.gz, snappy/build.go:222) through snaphash_tree in the library's DEFAULT configuration -- the
    archive is ONE stream, so the planner hands it to a host thread while the GPU takes the tree.  hashes.yaml is
    compared with the oracle's."""
    from oracle import oracle
    from snappy_amd import Context, synthetic
    base = shm_dir()
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
                "bytes": total, "ms": round(dt * 1e3, 1), "GiBps": round(total / GiB / dt, 2),
                "gpu_bytes": int(ex["gpu_bytes"]), "host_bytes": int(ex["host_bytes"]), "host_streams": int(ex["host_streams"]),
                "host_ms": round(ex["host_ms"], 1), "kernel_ms": round(st["kernel_ms"], 1),
                "model_vs_actual": model_vs_actual(ex),
                "cpu_port_serial_ms": round(dt_c * 1e3, 1),
                "parity": "hashes.yaml byte-identical to the oracle's", "best_of": 3}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def e2e_breakeven():
    """no call through the library may be slower than the loop it replaces.  Small trees on tmpfs through
    snaphash_tree in the DEFAULT configuration against the oracle's serial pass (the reference's one goroutine) over the
    same tree on the same box; the literal one-file helpers.Sha512sum call beside them.  All bit-exact."""
    import random
    from oracle import oracle
    from snappy_amd import Context
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import trees
    rng = random.Random(4)
    shapes = [
        ("24 files incl. one 1 MiB", [rng.randrange(100, 60000) for _ in range(23)] + [1 << 20] + [2048]),
        ("200 files incl. one 3 MiB", [rng.randrange(1000, 70000) for _ in range(199)] + [3 << 20] + [4096]),
        ("5000 x 8 KiB", [8192] * 5000 + [4096]),
        ("one 256 KiB file", [256 << 10] + [512]),
        ("3 MiB binary in a 10 MiB snap", [3 << 20] + [rng.randrange(20000, 120000) for _ in range(100)] + [4096]),
    ]
    rows = []
    base = shm_dir()
    with Context(flags=0) as c:  # ONE default ctx for all rows, as a process would hold it
        for name, sizes in shapes:
            tmp = tempfile.mkdtemp(prefix="snaphash_be_", dir=base)
            try:
                build, tar = trees.make_synthetic_tree(tmp, sizes)
                t0 = time.perf_counter()
                y = c.tree(build, tar)
                first = time.perf_counter() - t0
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    y = c.tree(build, tar)
                    ts.append(time.perf_counter() - t0)
                ex = c.stats_ex()
                to = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    want = oracle.hashes_yaml(build, tar)
                    to.append(time.perf_counter() - t0)
                if y != want:
                    raise SystemExit("PARITY FAILURE: breakeven leg (%s): hashes.yaml differs from the oracle's" % name)
                rows.append({"tree": name, "files": len(sizes) - 1, "bytes": int(sum(sizes)),
                             "library_default_ms": round(min(ts) * 1e3, 3), "first_call_ms": round(first * 1e3, 3),
                             "reference_serial_port_ms": round(min(to) * 1e3, 3),
                             "ratio": round(min(to) / min(ts), 2), "gpu_bytes": int(ex["gpu_bytes"]), "host_bytes": int(ex["host_bytes"]),
                             "model_vs_actual": model_vs_actual(ex),
                             "not_slower": bool(min(ts) <= min(to))})
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        # the literal call: helpers.Sha512sum(path) as a one-element batch
        tmp = tempfile.mkdtemp(prefix="snaphash_be_", dir=base)
        try:
            p = os.path.join(tmp, "one.bin")
            blob = os.urandom(256 << 10)
            open(p, "wb").write(blob)
            ts, to = [], []
            for _ in range(7):
                t0 = time.perf_counter()
                d = c.sha512_files([p])[0]
                ts.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                h = oracle.sha512sum(p)
                to.append(time.perf_counter() - t0)
            if d.hex() != h or h != hashlib.sha512(blob).hexdigest():
                raise SystemExit("PARITY FAILURE: breakeven leg: Sha512sum of one file")
            rows.append({"tree": "helpers.Sha512sum(one 256 KiB file), n = 1 batch", "files": 1, "bytes": 256 << 10,
                         "library_default_ms": round(min(ts) * 1e3, 3), "first_call_ms": round(ts[0] * 1e3, 3),
                         "reference_serial_port_ms": round(min(to) * 1e3, 3), "ratio": round(min(to) / min(ts), 2),
                         "gpu_bytes": int(c.stats_ex()["gpu_bytes"]), "host_bytes": int(c.stats_ex()["host_bytes"]),
                         "not_slower": bool(min(ts) <= min(to))})
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return {"what": "small trees (tmpfs) -> snaphash_tree, DEFAULT configuration, best of 5 after the first call, against the "
                    "oracle's serial writeHashes pass over the same tree (best of 3); hashes.yaml byte-identical in every row",
            "rows": rows, "every_row_not_slower": all(r["not_slower"] for r in rows)}


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


def build_cpu_baselines(out_gz, sha_rate_gibps):
    """What the fused build leg replaces, on this box's host cores, over the SAME tar stream (gunzipped back):
    (a) the reference's way, one core: compress/gzip level 9 (zlib level 9 here) over a 64 MiB sample, then the two
        SHA-512 passes it makes (archive, then every file again) at the C port's rate;
    (b) chunk-parallel zlib level 6 (1 MiB chunks, independent, pigz-style) on every core this job may keep busy."""
    import gzip
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    from snappy_amd import _lib
    tar = gzip.decompress(open(out_gz, "rb").read())
    sample = tar[:64 << 20]
    t0 = time.perf_counter()
    z9 = zlib.compress(sample, 9)
    dt9 = time.perf_counter() - t0
    rate9 = len(sample) / dt9                                  # B/s of tar stream, one core
    ratio9 = len(z9) / len(sample)
    sha_rate = sha_rate_gibps * GiB
    per_byte = 1.0 / rate9 + ratio9 / sha_rate + 1.0 / sha_rate  # compress + hash the archive + hash every file again
    cores = max(1, int(_lib.lib().snaphash_usable_cpus()))
    chunks = [tar[i:i + (1 << 20)] for i in range(0, len(tar), 1 << 20)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        outs = list(ex.map(lambda b: len(zlib.compress(b, 6)), chunks))
    dt6 = time.perf_counter() - t0
    return {"reference_way_one_core": {"GiBps_of_tree": round(1.0 / per_byte / GiB, 4), "cores": 1, "zlib9_MBps": round(rate9 / 1e6, 1),
                                       "ratio": round(ratio9, 4),
                                       "sample": "zlib level 9 over the first 64 MiB of the same tar stream (%.1f s) + two SHA-512 passes "
                                                 "at the C port's %.3f GiB/s (clickdeb/deb.go:271, snappy/build.go:222,241)" % (dt9, sha_rate_gibps)},
            "chunk_parallel_zlib6_all_cores": {"GiBps_of_tree": round(len(tar) / dt6 / GiB, 3), "cores": cores, "ratio": round(sum(outs) / len(tar), 4),
                                               "sample": "the whole tar stream (%d MiB) in independent 1 MiB chunks, zlib level 6, %d threads "
                                                         "(compression only: no SHA-512, no file I/O)" % (len(tar) >> 20, cores)}}


def e2e_build(ctx, sha_rate_gibps, total_mib=1024, ctx_kernels_only=None):
    """Rows f2 + f3: `Build`'s data step in one pass -- data.tar.gz (GPU DEFLATE) + archive digest + per-file SHA-512
    + hashes.yaml, every file read once -- on a compressible tree (Zipf-word text, 1 MiB files); the archive is read
    back with tarfile and the yaml compared with the oracle's over the tree and the archive just written.
    ctx: the library's DEFAULT configuration (what snaphash_init(NULL) gives a caller: the members' digests from host threads
    out of the staging buffer, the GPU compresses); ctx_kernels_only: the same pass with SNAPHASH_FLAG_GPU_ONLY (every member
    through the SHA-512 kernels beside the compressor), reported as `every_member_on_the_kernels`."""
    import tarfile
    from oracle import oracle
    from snappy_amd import synthetic
    base = shm_dir()
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
        kernels_only = None
        if ctx_kernels_only is not None:
            kb = None
            for _ in range(2):
                t0 = time.perf_counter()
                y2, dig2 = ctx_kernels_only.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
                dt2 = time.perf_counter() - t0
                if kb is None or dt2 < kb[0]:
                    kb = (dt2, ctx_kernels_only.stats(), ctx_kernels_only.targz_stats())
            if y2 != y or dig2 != dig:
                raise SystemExit("PARITY FAILURE: the fused build pass with every member on the kernels writes other bytes")
            kernels_only = {"ms": round(kb[0] * 1e3, 1), "sha512_kernel_ms": round(kb[1]["kernel_ms"], 1), "compressor_stream_ms": round(kb[2]["deflate_ms"], 1),
                            "what": "SNAPHASH_FLAG_GPU_ONLY: the members' SHA-512 kernels beside the compressor, whose pieces then run five rounds of "
                                    "workgroups instead of four (DESIGN.md sec. 9); same archive, same hashes.yaml", "best_of": 2}
        tf = tarfile.open(out, "r:gz")
        for k, m in enumerate(tf):
            if k >= 24:
                break
            if m.isreg() and tf.extractfile(m).read() != open(os.path.join(build, m.name[2:]), "rb").read():
                raise SystemExit("PARITY FAILURE: archive member %s differs from the file" % m.name)
        res = {"what": "on-disk tree (tmpfs, %d x 1 MiB Zipf-word text) -> snaphash_tar_create: tar + GPU DEFLATE + archive "
                       "SHA-512 + per-file SHA-512 + hashes.yaml, one read of every file" % total_mib,
               "tar_bytes": int(zs["tar_bytes"]), "gz_bytes": int(zs["gz_bytes"]), "ratio": round(zs["gz_bytes"] / zs["tar_bytes"], 4),
               "ms": round(dt * 1e3, 1), "GiBps_of_tree": round(zs["tar_bytes"] / GiB / dt, 2),
               "deflate_kernel_ms": round(zs["deflate_ms"], 1), "sha512_kernel_ms": round(st["kernel_ms"], 1),
               "members_hashed_by": "host threads out of the staging buffer (default configuration)" if st["kernel_ms"] == 0 else "the SHA-512 kernels",
               "every_member_on_the_kernels": kernels_only,
               "deflate_kernel": deflate_roofline(zs),
               "bound": "the serial SHA-512 of the archive on one host core (~183 ms for this stream) behind the arrival of the compressor's first piece; the compressor's stream just under it (DESIGN.md sec. 9)",
               "parity": "archive inflates to the tree (tarfile); archive digest = hashlib; hashes.yaml byte-identical to the oracle's",
               "best_of": 3}
        try:
            res["cpu_baseline"] = build_cpu_baselines(out, sha_rate_gibps)
        except Exception as e:  # noqa: BLE001
            res["cpu_baseline"] = {"error": repr(e)[:200]}
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def e2e_build_small():
    """The fused Build pass on packages as most snaps are -- many small files, a few long ones (sizes log-normal around
    20 KiB, up to 16 MiB; content: text and an executable's bytes) -- in the library's DEFAULT configuration, fresh ctx:
    first call and best of three, hashes.yaml against the oracle's, the archive's digest against hashlib.  A lone SHA-512 chain
    is 44 MB/s on the GPU: long members are hashed by host threads out of the staging buffer (`long_members`), and the
    producer's buffers are sized for the job (the first call)."""
    from oracle import oracle
    from snappy_amd import Context
    base = shm_dir()
    rng = np.random.default_rng(21)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
    text = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=(8 << 20) // 5 + 16) % 2000)[:8 << 20]
    pool = (text + open(sys.executable, "rb").read()) * 4
    rows = []
    for files in (300, 3000):
        sizes = np.minimum(16 << 20, np.maximum(1, rng.lognormal(np.log(20 << 10), 2.0, size=files))).astype(np.int64)
        tmp = tempfile.mkdtemp(prefix="snaphash_bsmall_", dir=base)
        try:
            build = os.path.join(tmp, "build")
            os.makedirs(os.path.join(build, "DEBIAN"))
            for i, sz in enumerate(sizes):
                d = os.path.join(build, "d%03d" % (i // 100))
                os.makedirs(d, exist_ok=True)
                off = int(rng.integers(0, len(pool) - int(sz) - 1))
                with open(os.path.join(d, "f%05d" % i), "wb") as f:
                    f.write(pool[off:off + int(sz)])
            out = os.path.join(tmp, "data.tar.gz")
            with Context(flags=0) as c:
                t0 = time.perf_counter()
                c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
                first = time.perf_counter() - t0
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    y, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
                    dt = time.perf_counter() - t0
                    if best is None or dt < best[0]:
                        best = (dt, c.targz_stats(), c.stats_ex())
            dt, zs, ex = best
            if hashlib.sha512(open(out, "rb").read()).digest() != dig or oracle.hashes_yaml(build, out) != y:
                raise SystemExit("PARITY FAILURE: the fused build pass of a small package disagrees with hashlib / the oracle")
            rows.append({"files": files, "MiB": round(int(sizes.sum()) / 2**20, 1), "largest_member_MiB": round(int(sizes.max()) / 2**20, 1),
                         "ms": round(dt * 1e3, 2), "first_call_ms": round(first * 1e3, 1), "ratio": round(zs["gz_bytes"] / zs["tar_bytes"], 4),
                         "long_members": int(ex["host_streams"]) - 1, "long_member_MiB": round((int(ex["host_bytes"]) - int(zs["gz_bytes"])) / 2**20, 1),
                         "gpu_only_floor_ms": round(int(sizes.max()) / 44e6 * 1e3, 1)})
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return {"what": "packages of 300 and 3 000 files (log-normal sizes around 20 KiB, up to 16 MiB) -> snaphash_tar_create with hashes.yaml, DEFAULT "
                    "configuration, fresh ctx: first call and best of three; gpu_only_floor_ms = what the largest member's SHA-512 chain alone takes on the GPU (44 MB/s)",
            "rows": rows, "parity": "hashes.yaml byte-identical to the oracle's, archive digest = hashlib, in every row"}


# ------------------------------------------------------------------------------------------
# BASELINE configs 1, 3, 5 on one GPU
# ------------------------------------------------------------------------------------------
def config_leg(name, device, kern, cpu_budget_s):
    """HBM-resident GPU-only pass (roofline), then the DEFAULT configuration from host memory, the CPU port beside it,
    sampled parity vs hashlib."""
    import torch
    from snappy_amd import Context, _lib, synthetic
    sizes = synthetic.config_sizes(name)
    n = len(sizes)
    total = int(sizes.sum())
    note = None
    free_host = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) * 1024
    cg = "/sys/fs/cgroup/memory.max"
    if os.path.exists(cg):
        lim = open(cg).read().strip()
        if lim.isdigit():
            free_host = min(free_host, int(lim))
    free_hbm = torch.cuda.mem_get_info()[0]
    if total + (8 << 30) > min(free_hbm, free_host - (24 << 30)):  # C3 needs 100 GiB of HBM and of host memory
        fit = max(1, int((min(free_hbm, free_host - (24 << 30)) - (8 << 30)) // int(sizes.max())))
        sizes = sizes[:min(n, fit)]
        note = "scaled to %d of %d streams: %d GiB of HBM / %d GiB of host memory free" % (len(sizes), n, free_hbm >> 30, free_host >> 30)
        n, total = len(sizes), int(sizes.sum())
    offsets, packed = synthetic.pack_offsets(sizes)
    findex = np.arange(n, dtype=np.uint64)
    out = {"streams": n, "bytes": total}
    if note:
        out["scaled"] = note
    rctx = Context(device=device, kernel=kern, stream=torch.cuda.current_stream().cuda_stream, flags=_lib.FLAG_GPU_ONLY)
    data = torch.empty(max(packed, 16), dtype=torch.uint8, device="cuda")
    rctx.fill_synthetic_device(data.data_ptr(), offsets, sizes, findex)
    slab = torch.zeros((n, 64), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    heavy = total / 44e6 / max(n, 1) > 1.0 or float(sizes.max()) / 44e6 > 2.0  # a pass of many seconds: one launch, no warm-up
    k_ms, wall, st = resident_pass(rctx, data.data_ptr(), offsets, sizes, slab.data_ptr(), 1 if heavy else 5, 0 if heavy else 1)
    want = slab.cpu().numpy()
    out["hbm_resident"] = {"ms": round(wall * 1e3, 3), "GiBps": round(total / GiB / wall, 3), "launches_timed": 1 if heavy else 5,
                           "roofline": roofline_of(k_ms, total, st, name, 1)}
    rctx.close()
    host = data.cpu().numpy()
    del data, slab
    torch.cuda.empty_cache()
    # sampled parity vs hashlib (OpenSSL), independent of both the library and the oracle
    rng = np.random.default_rng(3)
    sample = sorted(set([0, n - 1, int(np.argmax(sizes)), int(np.argmin(sizes))] + [int(x) for x in rng.integers(0, n, size=8)]))
    budget = 3 << 30
    checked = 0
    for i in sample:
        if budget < int(sizes[i]) and checked >= 2:
            continue
        budget -= int(sizes[i])
        if hashlib.sha512(host[int(offsets[i]):int(offsets[i]) + int(sizes[i])]).digest() != want[i].tobytes():
            raise SystemExit("PARITY FAILURE: config %s stream %d differs from hashlib.sha512" % (name, i))
        checked += 1
    out["parity"] = "%d sampled streams bit-exact vs hashlib.sha512; default-configuration vector identical to the GPU-only one" % checked
    # the DEFAULT configuration from host memory
    ptrs = (ctypes.c_void_p * n)(*[host.ctypes.data + int(o) for o in offsets])
    clens = (ctypes.c_uint64 * n)(*[int(x) for x in sizes])
    dig = ctypes.create_string_buffer(64 * n)
    with Context(device=device, flags=0) as c:
        best = None
        for _ in range(2 if total > (16 << 30) else 3):
            t0 = time.perf_counter()
            rc = _lib.lib().snaphash_sha512_buffers(c._h, ptrs, clens, n, dig)
            dt = time.perf_counter() - t0
            if rc:
                raise SystemExit("snaphash_sha512_buffers failed on config %s: %d" % (name, rc))
            if best is None or dt < best[0]:
                best = (dt, c.stats(), c.stats_ex())
        cpus = int(_lib.lib().snaphash_usable_cpus())
    if not np.array_equal(np.frombuffer(dig.raw, dtype=np.uint8).reshape(n, 64), want):
        raise SystemExit("PARITY FAILURE: config %s: default-configuration digests differ from the GPU-only pass" % name)
    dt, st, ex = best
    out["default_config"] = {"what": "host buffers -> snaphash_sha512_buffers, DEFAULT configuration (planned: planner.h)",
                             "ms": round(dt * 1e3, 2), "GiBps": round(total / GiB / dt, 2), "gpu_bytes": int(ex["gpu_bytes"]),
                             "host_bytes": int(ex["host_bytes"]), "host_streams": int(ex["host_streams"]), "host_ms": round(ex["host_ms"], 1),
                             "kernel_ms": round(st["kernel_ms"], 2), "h2d_ms": round(st["h2d_ms"], 2), "usable_cpus": cpus,
                             "model_vs_actual": model_vs_actual(ex)}
    order = list(np.argsort(-sizes.astype(np.int64))[:1]) + [int(x) for x in rng.permutation(n)[:4000]] if name == "C5" else None
    out["cpu_port"] = cpu_port_sample(host, offsets, sizes, cpu_budget_s, order)
    if name == "C5":
        head = float(sizes.max())
        out["note"] = ("bound by its %d MiB head: ONE stream, %.2f s on one host core at the measured host rate and %.1f s on the GPU; "
                       "no number of GPUs shortens it, so C5 cannot scale with N (DESIGN.md sec. 6)" %
                       (int(head) >> 20, head / 1.4e9, head / 44e6))
    if name == "C3":
        out["note"] = ("100 streams x 44 MB/s = 4.4 GB/s is all the GPU can do for 100 SHA-512 chains; the planner gives every "
                       "stream to a host thread (%d usable cores here), the GPU hashes 0 bytes in the default configuration" % cpus)
    del host
    return out


# ------------------------------------------------------------------------------------------
# --gpus N > 1 without an outer launcher: the parent starts the ranks itself
# ------------------------------------------------------------------------------------------
def self_launch(args):
    """`python3 bench.py --gpus N` by itself (no torch.distributed.run around it): N rank processes as CHILDREN of this
    one, the rendezvous variables in their environment, rank 0's stdout (the ONE JSON line) relayed, every other rank's
    stdout sent to stderr.  Runs before torch, snappy_amd or anything else that could touch a GPU is imported here, and
    never replaces this process (no exec): the parent only waits.  If a rank fails the others are ended (their exact
    PIDs) and the parent exits with that rank's code."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        # (OMP_NUM_THREADS: torch.distributed.run gives its ranks 1 unless the caller says otherwise, and for a reason that was
        # measured here -- every rank's own OpenMP pool, as wide as the box (256), spins after each small tensor operation of the
        # collectives and burns the job's CPU quota: two ranks took 419 ms a step instead of 202, gpurun_out/r5w)
        env.setdefault("OMP_NUM_THREADS", "1")
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "SNAPHASH_BENCH_SELF_LAUNCHED": "1",
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import signal
    import threading

    def pass_on(signum, frame):  # the parent is told to stop (a driver's time limit): so are the ranks, by their exact PIDs
        for q in procs:
            if q.poll() is None:
                q.terminate()
        raise SystemExit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, pass_on)
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    live = set(range(n))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        time.sleep(0.05)
    if failed is not None:
        for r in sorted(live):
            procs[r].terminate()
        deadline = time.time() + 20
        for r in sorted(live):
            try:
                procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    reader.join(timeout=10)
    text = (out0[0] if out0 else b"").decode(errors="replace")
    sys.stdout.write(text)
    sys.stdout.flush()
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with %d; the other ranks were ended\n" % failed)
        return failed[1] if 0 < failed[1] < 256 else 1
    if not any(l.startswith("{") for l in text.splitlines()):
        sys.stderr.write("bench.py: the ranks ended without a JSON line from rank 0\n")
        return 1
    return 0


# ------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # nothing above this line has imported torch or the library
    # stdout carries ONE JSON line and nothing else: whatever the libraries below print there (RCCL's version banner
    # at the first collective, for one) goes to stderr
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    t_start = time.perf_counter()
    import torch
    import torch.distributed as dist
    from snappy_amd import Context, _lib, synthetic
    from snappy_amd.sharded import ShardedTree

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:  # (only an environment that SETS WORLD_SIZE=1 gets here: unset, the ranks are self-launched)
            raise SystemExit("--gpus %d under WORLD_SIZE=1: unset WORLD_SIZE (bench.py then starts its own ranks) or launch %d ranks" % (args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback exists)"
    # SNAPHASH_BENCH_SAME_GPU=1: every rank on GPU 0, control plane and digest gather over gloo (RCCL refuses two ranks on
    # one device): how a 1-GPU box rehearses the N > 1 path.  SNAPHASH_BENCH_FORCE_DIST=1: the RCCL path with one rank.
    same_gpu = os.environ.get("SNAPHASH_BENCH_SAME_GPU") == "1"
    # LOCAL_RANK names the rank's GPU; a launcher that hands every rank ONE visible device (ordinal 0) is accommodated
    visible = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    self_launched = os.environ.get("SNAPHASH_BENCH_SELF_LAUNCHED") == "1"  # (our own children all see what the parent saw)
    if not same_gpu and world > 1 and visible < local_world and (visible != 1 or self_launched):
        # (visible == 1: a launcher that hands every rank ONE device of its own; anything else would put two ranks on one
        # GPU, which RCCL refuses later and less legibly)
        raise SystemExit("%d ranks on this node but %d GPUs visible: one rank per GPU is the contract (SNAPHASH_BENCH_SAME_GPU=1 "
                         "rehearses N ranks on one GPU over gloo)" % (local_world, visible))
    device = 0 if same_gpu else local_rank % max(1, visible)
    torch.cuda.set_device(device)
    force_dist = os.environ.get("SNAPHASH_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force_dist
    coll_device = None if same_gpu else "cuda"
    backend = None
    if use_dist and "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(1)  # (a launcher that did not say: N ranks x an OpenMP pool as wide as the box oversubscribe the node; see self_launch)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if same_gpu else "nccl"
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    # what moves the digest slabs, under its own name: backend "nccl" IS RCCL on ROCm; the one-GPU rehearsal runs over gloo
    coll_name = {"nccl": "RCCL", "gloo": "gloo (CPU tensors: the same-GPU rehearsal)", None: "none"}[backend]

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_obj(obj):
        if not use_dist:
            return [obj]
        objs = [None] * world
        dist.all_gather_object(objs, obj)
        return objs

    def bcast_obj(obj):
        if not use_dist:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    # One process per GPU: a rank's library sees the whole job's CPU allowance (affinity mask, cgroup quota) and would size its
    # fill pool for it.  When the allowance divided by the ranks is below what an engine takes by itself (12 fill threads),
    # every rank is told its share (the in-library form, one ctx for all devices, divides by itself: snaphash_init).
    if world > 1:
        share = int(_lib.lib().snaphash_usable_cpus()) // world
        if share < 12:
            os.environ.setdefault("SNAPHASH_COPY_THREADS", str(max(2, share)))
    kern = {"auto": _lib.KERNEL_AUTO, "wide": _lib.KERNEL_WIDE, "split": _lib.KERNEL_SPLIT, "pair": _lib.KERNEL_PAIR,
            "quad": getattr(_lib, "KERNEL_QUAD", 4)}[args.kernel]
    numa_note = bind_to_gpu_node(device)
    sizes = synthetic.config_sizes(args.workload)
    n_files = len(sizes) - 1           # the last size is the archive stand-in
    total_bytes = int(sizes.sum())

    # ---- the ONE on-disk tree (tmpfs).  Rank 0 lays out directories and empty files of the right size; every rank then
    # plans (the walk sees the sizes) and writes the content of ITS members, so their page-cache pages sit on its socket.
    tmp = None
    tree_home = None
    if rank == 0:
        need = total_bytes + (4 << 30)
        # tmpfs first (the reads then cost what the page cache costs, which is what a build's second read of its files
        # meets: clickdeb/deb.go:285-341 has just read them); a node without room there gets the temp dir's file system
        # (after the write the pages are in the page cache all the same)
        homes = [d for d in (shm_dir(), tempfile.gettempdir(), os.path.join(ROOT, "gpurun_out")) if d and os.path.isdir(d)]
        base = next((d for d in homes if shutil.disk_usage(d).free >= need), None)
        if base is None:
            raise SystemExit("no room for the %d MiB tree in any of %s" % (total_bytes >> 20, homes))
        tree_home = base
        tmp = tempfile.mkdtemp(prefix="snaphash_bench_", dir=base)
        build0 = os.path.join(tmp, "build")
        os.makedirs(build0)
        made = set()
        for i in range(n_files):
            p = os.path.join(build0, synthetic.file_name(i))
            d = os.path.dirname(p)
            if d not in made:
                os.makedirs(d, exist_ok=True)
                made.add(d)
            with open(p, "wb") as f:
                f.truncate(int(sizes[i]))
        with open(os.path.join(tmp, "data.tar.gz"), "wb") as f:
            f.truncate(int(sizes[n_files]))
    tmp = bcast_obj(tmp)
    build, tar = os.path.join(tmp, "build"), os.path.join(tmp, "data.tar.gz")
    line = None
    try:
        if use_dist:
            dist.barrier()
        rstream = torch.cuda.current_stream().cuda_stream
        rctx = Context(device=device, kernel=kern, stream=rstream, flags=_lib.FLAG_GPU_ONLY)  # the resident (roofline) pass: torch's stream
        ectx = Context(device=device, kernel=kern, flags=_lib.FLAG_GPU_ONLY)                    # the timed pass: own stream, own staging
        # N > 1: the ranks SHARE the walk (ABI 5 snaphash_shard_list / _plan_from): each walks every world-th entry of the root,
        # two small all-gathers move the listings, every rank rebuilds the same records -- instead of `world` full walks
        share = ({"device": coll_device, "force": force_dist} if use_dist and (world > 1 or force_dist) and os.environ.get("SNAPHASH_BENCH_FULL_WALKS") != "1"
                 else None)
        plan = ShardedTree(build, tar, rank, world, share_walk=share)
        my_paths = plan.paths()
        my_index = np.array([file_index_of(p, n_files) for p in my_paths], dtype=np.uint64)
        my_lens = np.ascontiguousarray(sizes[my_index.astype(np.int64)]) if len(my_paths) else np.zeros(0, dtype=np.uint64)
        my_off, my_total = synthetic.pack_offsets(my_lens)
        data = torch.empty(max(my_total, 16), dtype=torch.uint8, device="cuda")
        rctx.fill_synthetic_device(data.data_ptr(), my_off, my_lens, my_index)
        torch.cuda.synchronize()
        host = data.cpu().numpy()  # the same bytes in this rank's host memory: written to its files, and the source of the buffers leg
        for k, p in enumerate(my_paths):
            host[int(my_off[k]):int(my_off[k]) + int(my_lens[k])].tofile(p)

        # ---- (i) kernel-resident: this rank's streams already in HBM -> roofline ---------------------------------
        slab = torch.zeros((max(plan.rows, 1), 64), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        k_ms, r_wall, r_st = resident_pass(rctx, data.data_ptr(), my_off, my_lens, slab.data_ptr(), args.resident_steps, 3)
        my_want = slab[:len(my_paths)].cpu().numpy()
        r_wall = allmax(r_wall)
        del data
        torch.cuda.empty_cache()

        # ---- (ii) end to end: the timed region ---------------------------------------------------------------------
        last = {}

        def step(c=None, key="yaml"):
            c = c or ectx
            if world == 1 and not force_dist:
                last[key] = c.tree(build, tar)
                return
            with ShardedTree(build, tar, rank, world, share_walk=share) as st:
                mine = st.hash(c)
                slabs = st.gather(mine, device=coll_device)
                if rank == 0:
                    last[key] = st.emit(slabs)
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        step_ms = []
        for _ in range(args.steps):
            ts = time.perf_counter()
            step()
            step_ms.append((time.perf_counter() - ts) * 1e3)  # (this rank's own view of each step: reported beside the mean)
        fence()
        elapsed = allmax(time.perf_counter() - t0)
        e_st = ectx.stats()
        per_rank_step = gather_obj({"streams": len(my_paths), "bytes": int(my_lens.sum()), "h2d_ms": round(e_st["h2d_ms"], 2),
                                    "kernel_ms": round(e_st["kernel_ms"], 2), "launches": int(e_st["launches"])})
        ms_step = elapsed / args.steps * 1e3
        value = total_bytes / GiB / (elapsed / args.steps)
        # who took part: every rank's GPU by its PCI bus id (snaphash_get_engine_info), N distinct ones or the line says so
        e_info = ectx.engine_info(0)
        rank_ids = gather_obj({"rank": rank, "local_rank": local_rank, "device": device, "pci_bus_id": e_info["pci_bus_id"],
                               "numa_node": e_info["numa_node"], "pid": os.getpid(), "host": os.uname().nodename})

        # ---- parity of the timed path, outside the timed region ----------------------------------------------------
        parity, cpu, oracle_yaml = None, None, None
        full_legs = rank == 0 and world == 1 and not force_dist and (args.legs == "full" or (args.legs == "auto" and args.workload == "C2"))
        if rank == 0:
            y = last["yaml"]
            rng = np.random.default_rng(1)
            sample = sorted(set([0, n_files - 1] + [int(x) for x in rng.integers(0, n_files, size=14)]))
            names = [synthetic.file_name(i) for i in sample]
            hashlib_check([os.path.join(build, nm) for nm in names], [yaml_digest_of(y, nm) for nm in names], "timed path")
            parity = {"hashlib_sample": "%d files of the timed pass's hashes.yaml bit-exact vs hashlib.sha512" % len(sample),
                      "yaml_bytes": len(y), "records": y.count(b"- name: "), "sha512_of_yaml": hashlib.sha512(y).hexdigest()[:32]}
            if world > 1 or force_dist:  # the sharded YAML against the single-GPU pass over the same tree
                y1 = ectx.tree(build, tar)
                if y1 != y:
                    raise SystemExit("PARITY FAILURE: the sharded pass's hashes.yaml differs from the single-GPU snaphash_tree's")
                parity["single_gpu"] = "hashes.yaml of the %d-rank pass byte-identical to the single-GPU snaphash_tree's" % world
            if args.cpu_seconds > 0:
                # at every N: the oracle's own serial pass over the same on-disk tree is both the checker of the timed path's
                # hashes.yaml and the cpu_baseline (rank 0, outside the timed region; the other ranks wait at the next fence)
                from oracle import oracle
                t0 = time.perf_counter()
                oracle_yaml = oracle.hashes_yaml(build, tar)
                dt_c = time.perf_counter() - t0
                if oracle_yaml != y:
                    raise SystemExit("PARITY FAILURE: hashes.yaml differs from the oracle's")
                parity["result"] = "hashes.yaml%s byte-identical to the oracle's (%d records)" % (
                    " of the %d-rank pass" % world if world > 1 else "", y.count(b"- name: "))
                cpu = {"value": round(total_bytes / GiB / dt_c, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
                       "sample": "the oracle's whole writeHashes pass (walk, 32 KiB reads, SHA-512, YAML) over the same on-disk "
                                 "tree of %d files / %.0f MiB: %.1f s on one core, as the reference's single goroutine would run it" %
                                 (n_files + 1, total_bytes / 2**20, dt_c),
                       "host_cores_visible": os.cpu_count(), "usable_cpus": int(_lib.lib().snaphash_usable_cpus())}
                if world == 1:
                    cpu["all_cores"] = cpu_pool_leg(sizes, 3.0)
            elif "result" not in parity:
                parity["result"] = parity.get("single_gpu", "hashlib sample only") + " (--cpu-seconds 0: no oracle pass)"

        # ---- end_to_end legs ------------------------------------------------------------------------------------------
        end_to_end, configs = {}, None

        def leg(store, name, fn):
            # an auxiliary leg that cannot run here (no room in /dev/shm, ...) is recorded, it never costs the headline
            # line; a PARITY FAILURE is a SystemExit and still ends the run
            t0 = time.perf_counter()
            try:
                r = fn()
                if isinstance(r, dict):
                    r["leg_seconds"] = round(time.perf_counter() - t0, 1)
                store[name] = r
            except Exception as e:  # noqa: BLE001
                store[name] = {"error": repr(e)[:300]}
        if args.legs != "off":
            bctx = Context(device=device, kernel=kern, flags=_lib.FLAG_GPU_ONLY)
            r = e2e_buffers_sharded(bctx, host, my_off, my_lens, my_want, total_bytes, world, fence, allmax, gather_obj)
            bctx.close()
            r["source_placement"] = numa_note
            end_to_end["buffers_sharded"] = r
            if use_dist and args.legs == "full":  # (--legs full: an optional collective leg is not worth a rank's failure costing the line)
                # The same sharded pass in the library's DEFAULT configuration: every rank plans its share (planner.h) and the
                # host cores beside its staging fill take what shortens it -- what the whole node does when its cores may
                # help.  Never `value`: the headline keeps every byte on the GPUs.
                try:
                    dctx = Context(device=device, kernel=kern, flags=0)
                    best = None
                    for _ in range(3):
                        fence()
                        t0 = time.perf_counter()
                        step(dctx, "yaml_default")
                        fence()
                        dt = allmax(time.perf_counter() - t0)
                        best = dt if best is None else min(best, dt)
                    ex = dctx.stats_ex()
                    dctx.close()
                    per = gather_obj({"gpu_bytes": int(ex["gpu_bytes"]), "host_bytes": int(ex["host_bytes"]), "host_streams": int(ex["host_streams"])})
                    if rank == 0:
                        if last.get("yaml_default") != last["yaml"]:
                            raise SystemExit("PARITY FAILURE: the default-configuration sharded pass writes another hashes.yaml")
                        end_to_end["tree_sharded_default"] = {
                            "what": "the timed pass with every rank in the DEFAULT configuration (planned; host cores take a share)",
                            "ms": round(best * 1e3, 2), "GiBps": round(total_bytes / GiB / best, 2), "per_rank": per,
                            "usable_cpus_per_rank": int(_lib.lib().snaphash_usable_cpus()),
                            "parity": "hashes.yaml byte-identical to the GPU-only pass", "best_of": 3}
                except SystemExit:
                    raise
                except Exception as e:  # noqa: BLE001
                    if rank == 0:
                        end_to_end["tree_sharded_default"] = {"error": repr(e)[:300]}
        if full_legs:
            verify_ms = None
            t0 = time.perf_counter()
            if ectx.verify(build, last["yaml"], tar) is not None:
                raise SystemExit("Verify reports a mismatch on the tree just hashed")
            verify_ms = (time.perf_counter() - t0) * 1e3
            end_to_end["tree"] = {"what": "the timed pass itself (value): on-disk tree (tmpfs) -> snaphash_tree -> hashes.yaml",
                                  "files": n_files + 1, "bytes": total_bytes, "ms": round(ms_step, 2), "GiBps": round(value, 2),
                                  "h2d_ms": round(e_st["h2d_ms"], 2), "kernel_ms": round(e_st["kernel_ms"], 2),
                                  "verify_ms": round(verify_ms, 2), "yaml_bytes": len(last["yaml"])}
            # the shard entry points with one rank must cost what snaphash_tree costs (the N > 1 value is built from them)
            def tree_world1():
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    with ShardedTree(build, tar, 0, 1) as st1:
                        y1 = st1.emit(st1.gather(st1.hash(ectx)))
                    ts.append(time.perf_counter() - t0)
                if y1 != last["yaml"]:
                    raise SystemExit("PARITY FAILURE: snaphash_shard_* with one rank differs from snaphash_tree")
                return {"what": "snaphash_shard_plan/_hash/_emit with world = 1 over the same tree (what N > 1 times, less the all-gather)",
                        "ms": round(min(ts) * 1e3, 2), "GiBps": round(total_bytes / GiB / min(ts), 2), "best_of": 3}
            leg(end_to_end, "tree_sharded_world1", tree_world1)
            leg(end_to_end, "tree_default", lambda: e2e_tree_default(build, tar, total_bytes, last["yaml"], device))
        del host
        shutil.rmtree(tmp, ignore_errors=True) if rank == 0 and not use_dist else None
        if full_legs:
            leg(end_to_end, "package", e2e_package)
            sha_rate = cpu["value"] if cpu else 0.47
            def build_leg():
                with Context(device=device, kernel=kern, flags=0) as bctx:  # the defaults a cgo caller gets
                    return e2e_build(bctx, sha_rate, ctx_kernels_only=ectx)
            leg(end_to_end, "build", build_leg)
            leg(end_to_end, "build_small", e2e_build_small)
            leg(end_to_end, "breakeven", e2e_breakeven)
        ectx.close()
        rctx.close()
        if full_legs:
            configs = {}
            for name in ("C1", "C5", "C3"):
                leg(configs, name, lambda nm=name: config_leg(nm, device, kern, 3.0 if args.cpu_seconds > 0 else 0.0))

        if rank == 0:
            distinct = len(set((r["host"], r["pci_bus_id"]) for r in rank_ids))
            wl = "%s: ONE on-disk tree (tmpfs) of %d files%s -> hashes.yaml, sharded over %d rank(s) on %d distinct GPU(s)%s" % (
                args.workload, n_files + 1, " (10 000 x 1 MiB + the 1 MiB archive stand-in)" if args.workload == "C2" else "",
                world, distinct, ", %s all-gather of the digest slabs" % coll_name if use_dist else "")
            ranks = {"world_size": dist.get_world_size() if use_dist else 1, "backend": backend, "collective": coll_name,
                     "launcher": "bench.py's own child processes" if os.environ.get("SNAPHASH_BENCH_SELF_LAUNCHED") == "1"
                                 else ("outer launcher (torch.distributed.run or alike)" if "RANK" in os.environ else "single process"),
                     "gpus": rank_ids, "distinct_gpus": distinct,
                     "one_gpu_per_rank": bool(distinct == world)}
            if distinct != world:
                ranks["note"] = ("%d ranks SHARE %d GPU(s): a rehearsal of the N > 1 path (SNAPHASH_BENCH_SAME_GPU), not a "
                                 "multi-GPU measurement -- `value` here says nothing about scaling" % (world, distinct))
            line = {
                "metric": "GiB/s hashed (whole node), bit-exact hashes.yaml, 10k x 1 MiB tree" if args.workload == "C2"
                          else "GiB/s hashed (whole node), bit-exact hashes.yaml, workload %s" % args.workload,
                "value": round(value, 3), "unit": "GiB/s",
                "value_kind": "end_to_end (SURVEY sec. 8d-ii): walk + pread + pinned staging + H2D over PCIe + HIP kernels + "
                              "hashes.yaml, every byte hashed by the kernels; " +
                              ("snaphash_tree on one GPU" if world == 1 and not force_dist else
                               "the ranks share the walk (every world-th entry of the root each, the listings all-gathered), every rank its LPT share of the ONE tree (snaphash_shard_*), one all-gather of the slabs over %s, rank 0 writes the YAML" % coll_name) +
                              ".  The HBM-resident rate is `hbm_resident` / `roofline`, never `value`",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": wl, "files": n_files + 1, "bytes": total_bytes, "kernel": _lib.KERNEL_NAMES.get(r_st["kernel_used"], "?"),
                           "ranks": ranks, "per_rank": per_rank_step, "tree_placement": numa_note, "tree_home": tree_home,
                           "step_ms_rank0": {"min": round(min(step_ms), 2), "median": round(sorted(step_ms)[len(step_ms) // 2], 2), "max": round(max(step_ms), 2)}},
                "roofline": roofline_of(k_ms, int(my_lens.sum()), r_st, args.workload, world, streams=len(my_paths)),
                "hbm_resident": {"what": "the same streams already in HBM (snaphash_sha512_device), every rank its share; wall per pass, MAX over ranks",
                                 "ms_per_pass": round(r_wall * 1e3, 4), "GiBps": round(total_bytes / GiB / r_wall, 3),
                                 "launches_timed": args.resident_steps,
                                 "note": "stream-count-bound: 10 001 streams advance no faster on 8 GPUs than on one, so this rate is "
                                         "flat in N by construction (DESIGN.md sec. 5)"},
                "parity": parity,
                "ceilings": {"hbm_GBps": HBM_PEAK_GBPS, "valu_saturated_GBps_measured": VALU_SATURATED_GBPS,
                             "per_stream_MBps_here": round(int(my_lens.sum()) / (k_ms * 1e-3) / 1e6 / max(len(my_paths), 1), 2),
                             "pcie_link_GBps_measured": 55.3, "source": "profiles/ (regime sweep, saturated-launch PMC, r04_shard_probe)"},
            }
            if end_to_end:
                line["end_to_end"] = end_to_end
            if configs:
                line["configs"] = configs
            if cpu is not None:
                line["cpu_baseline"] = cpu
    finally:
        if use_dist:
            try:
                dist.barrier()
            except Exception:  # noqa: BLE001
                pass
        if rank == 0 and tmp:
            shutil.rmtree(tmp, ignore_errors=True)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and line is not None:
        # ---- N > 1: the same tree through ONE process and the C ABI's device list (what the Go caller reaches over
        # cgo): in-library LPT shards, one NUMA-placed staging engine per GPU, single-process RCCL gather.  Run when
        # the ranks have let go of their GPUs, in a child process with a time limit: the line never depends on it. ----
        if world > 1 and not same_gpu and os.environ.get("SNAPHASH_BENCH_NO_INLIB") != "1":
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
            line["in_library_multi_gpu"] = inlib
        td = line.get("end_to_end", {}).get("tree_default") if isinstance(line.get("end_to_end"), dict) else None
        if isinstance(td, dict) and "ms" in td:  # beside `value` (every byte through the kernels): what snaphash_init(NULL) does with the same tree
            line["default_configuration"] = {"ms": td["ms"], "GiBps": td["GiBps"], "host_bytes": td.get("host_bytes"), "gpu_bytes": td.get("gpu_bytes"),
                                             "what": "end_to_end.tree_default: the same tree -> hashes.yaml as the library plans it by default (a share of the files on host "
                                                     "threads beside the PCIe link, eight streams a thread in AVX-512 lanes); never `value`"}
        # the planner's conscience in one place: every planned leg's prediction against what it took, and which missed by > 25 %
        def mva_rows(node, path, out):
            if isinstance(node, dict):
                if "model_vs_actual" in node and isinstance(node["model_vs_actual"], dict):
                    m = node["model_vs_actual"]
                    out[path or "."] = {"call": m.get("call"), "off": m.get("off_by_more_than_25_pct", [])}
                for k, v in node.items():
                    if k != "model_vs_actual":
                        mva_rows(v, (path + "." if path else "") + str(k), out)
            elif isinstance(node, list):
                for i, v in enumerate(node):
                    mva_rows(v, "%s[%s]" % (path, v.get("tree", i) if isinstance(v, dict) else i), out)
        rows = {}
        mva_rows(line.get("end_to_end"), "end_to_end", rows)
        mva_rows(line.get("configs"), "configs", rows)
        if rows:
            line["planner"] = {"what": "snaphash_stats_ex (ABI 5): the plan's modelled makespan beside the measured one for every leg that "
                                       "ran in the default (planned) configuration; the model's link and fill rates are calibrated on "
                                       "this box (snaphash_get_calib), see end_to_end.tree_default.plan_model",
                               "rows": rows, "rows_off_by_more_than_25_pct": sorted(k for k, v in rows.items() if v["off"])}
        line["bench_seconds"] = round(time.perf_counter() - t_start, 1)
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()


if __name__ == "__main__":
    main()
