"""The planner (snappy_amd/csrc/planner.cpp, ABI 4 snaphash_plan_streams): which streams of a call the HIP kernels hash and
which the library's own host SHA-512.  Host-only: no GPU is touched here; what the plan does on the MI355X is
tests/test_gpu_api2.py::test_default_configuration_* and bench.py's end_to_end.breakeven leg.

The seam it protects is helpers.Sha512sum and its loop in writeHashes (reference helpers/helpers.go:187-201,
snappy/build.go:222,241): ONE goroutine at ~0.5 GB/s.  A lone SHA-512 stream advances at ~44 MB/s on the GPU, so any
call dominated by one stream must not run there."""
import os

import pytest

MiB = 1 << 20
REF_RATE = 0.47 * 2**30  # the reference's single goroutine as the C port measures it on the GPU box (BENCH_r03 cpu_baseline)


def _plan(lens, **kw):
    from snappy_amd import _lib
    kw.setdefault("cpus", 16)
    return _lib.plan_streams(lens, **kw)


def _makespan(r):
    return max(r["gpu_seconds"], r["host_seconds"])


# VERDICT r3's four breakeven shapes, and the literal one-file helpers.Sha512sum call
BREAKEVEN = {
    "one 256 KiB file": [256 << 10],
    "24 files incl. one 1 MiB": [1000, 50000, 200000, 3, 4096] * 4 + [1 * MiB, 0, 77, 12345],
    "200 files incl. one 3 MiB": [3 * MiB] + [35000] * 199,
    "5000 x 8 KiB": [8192] * 5000,
    "3 MiB binary in a 10 MiB snap": [3 * MiB] + [70000] * 100,
}


@pytest.mark.parametrize("shape", sorted(BREAKEVEN))
@pytest.mark.parametrize("from_files", [0, 1])
def test_no_call_is_modelled_slower_than_the_reference_loop(built_lib, shape, from_files):
    lens = BREAKEVEN[shape]
    on_host, r = _plan(lens, from_files=from_files)
    serial = sum(lens) / REF_RATE + (8e-6 * len(lens) if from_files else 0)
    assert _makespan(r) <= serial, (shape, r, serial)
    # and the stream that would take longest on the GPU is not there
    longest = max(range(len(lens)), key=lambda i: lens[i])
    if lens[longest] / 44e6 > serial:
        assert on_host[longest] == 1


def test_the_one_file_call_starts_no_gpu_part_and_one_thread(built_lib):
    on_host, r = _plan([256 << 10])
    assert on_host == [1] and r["gpu_seconds"] == 0 and r["host_threads"] == 1
    on_host, r = _plan([77])
    assert on_host == [1] and r["host_threads"] == 1


def test_many_similar_streams_stay_mostly_on_the_gpu(built_lib):
    """BASELINE config 2: the GPU part is bound by its PCIe link, so the cores the staging fill leaves free take a share
    -- and only a share: the kernels keep what the link carries."""
    lens = [MiB] * 10001
    on_host, r = _plan(lens)
    assert 0 < r["host_bytes"] < 0.35 * sum(lens)
    assert r["gpu_seconds"] < 10001 * MiB / 54e9 and abs(r["gpu_seconds"] - r["host_seconds"]) < 0.02
    # a batch the link is NOT the bound of (1 024 x 1 MiB: the streams' own 24 ms) is left alone
    on_host, r = _plan([MiB] * 1024)
    assert r["host_streams"] == 0


def test_the_archive_beside_its_tree_moves_and_nothing_else(built_lib):
    lens = [512 * MiB] + [MiB] * 512
    on_host, r = _plan(lens, from_files=1)
    assert on_host[0] == 1 and sum(on_host) == 1
    assert r["host_threads"] == 1


def test_few_huge_streams_go_to_the_host_whole(built_lib):
    """BASELINE config 3 (100 x 1 GiB): 100 streams x 44 MB/s = 4.4 GB/s on the GPU, 16 cores x 1.4 GB/s on the host."""
    on_host, r = _plan([1 << 30] * 100)
    assert sum(on_host) == 100 and r["gpu_seconds"] == 0 and r["host_threads"] == 16
    on_host, r = _plan([1 << 30] * 100, cpus=128)
    assert r["host_threads"] == 100  # a thread per stream at most


def test_explicit_thread_count_and_more_cores(built_lib):
    lens = [MiB] * 10001
    _, r4 = _plan(lens, host_threads=4)
    _, r8 = _plan(lens, host_threads=8)
    _, r32 = _plan(lens, host_threads=32)
    # a link-bound batch is split only for a modelled 10 % or more (profiles/r04_default_probe.txt): four threads add 9 %
    assert r4["host_bytes"] == 0 and r8["host_threads"] == 8 and r32["host_threads"] == 32 and r32["host_bytes"] > r8["host_bytes"] > 0
    _, one = _plan(lens, cpus=1)  # nothing to spare beside the staging fill: one thread helps a little or not at all
    assert one["host_bytes"] < 0.05 * sum(lens)


def test_many_small_files_and_a_few_big_ones_are_split_not_sent_to_the_host_whole(built_lib):
    """BASELINE config 5 as an on-disk tree (100 000 Zipf files): the GPU part is bound by its longest stream, not by its
    link, so it needs few cores and the big files get them -- 0.26 s on the GPU box, where sending every stream to the 16
    host threads (what a per-file cost of 4 us made the planner believe in) took 0.31-0.34 (profiles/r04_c5_on_disk_tree.txt)."""
    from snappy_amd import synthetic
    lens = [int(x) for x in synthetic.config_sizes("C5")]
    for from_files in (1, 0):
        on_host, r = _plan(lens, from_files=from_files)
        assert 10 <= r["host_streams"] <= 200 and r["host_bytes"] > 0.25 * sum(lens)  # the head and its like, nothing else
        assert 3 <= r["host_threads"] <= 14 and r["gpu_seconds"] > 0
        head = max(range(len(lens)), key=lambda i: lens[i])
        assert on_host[head] == 1 and abs(r["host_seconds"] - lens[head] / 1.4e9) < 0.03  # the floor: the head on one core


def test_a_link_bound_tree_gives_up_what_the_cores_can_hash_while_they_feed_it(built_lib):
    """Config 2 from files: the fill threads keep their count (they work in bursts), host threads come on top, and the
    cores' total binds: a seventh of the bytes at 16 cores (175 ms against 198 GPU-only, profiles/r04_default_probe.txt),
    none at 8, most of them at 128."""
    lens = [MiB] * 10001
    _, r16 = _plan(lens, from_files=1, cpus=16)
    assert 0.08 * sum(lens) < r16["host_bytes"] < 0.20 * sum(lens) and 4 <= r16["host_threads"] <= 10
    assert r16["gpu_seconds"] < 0.9 * 10001 * MiB / 54e9
    _, r8 = _plan(lens, from_files=1, cpus=8)
    assert r8["host_bytes"] == 0
    _, r128 = _plan(lens, from_files=1, cpus=128)
    assert r128["host_bytes"] > 0.5 * sum(lens) and r128["host_threads"] > 64
    # a rank's shard of it (1 250 x 1 MiB) is bound by its streams' own 24 ms: left alone
    assert _plan([MiB] * 1250, from_files=1)[1]["host_streams"] == 0


def test_small_files_stay_with_the_gpu_part(built_lib):
    """A small file costs its open + close more than its bytes, and that is one lock per process whoever pays it
    (profiles/r04_openat_probe.txt): beside a GPU part it does not move to a host thread -- 100 000 x 8 KiB took 148 ms with a
    quarter of them on host threads, 115 ms whole (profiles/r04_small_files_tree.txt).  From memory there is no such cost."""
    for lens in ([8192] * 100000, [65536] * 50000, [8192] * 5000):
        on_host, r = _plan(lens, from_files=1)
        assert r["host_streams"] == 0 and r["gpu_seconds"] > 0, (len(lens), r)
    _, r = _plan([100 << 10] * 20000, from_files=0)
    assert r["host_streams"] > 0
    # the big members of such a tree still move, and nothing below 256 KiB with them
    lens = [8192] * 20000 + [8 * MiB] * 6 + [200 << 10] * 50
    on_host, r = _plan(lens, from_files=1)
    assert all(on_host[i] == 0 for i in range(len(lens)) if lens[i] < (256 << 10))
    assert sum(on_host) >= 1 and all(lens[i] == 8 * MiB for i in range(len(lens)) if on_host[i])
    # and a batch with no GPU part left at all (a few dozen members) still goes to the host whole
    on_host, r = _plan(BREAKEVEN["24 files incl. one 1 MiB"], from_files=1)
    assert sum(on_host) == 24 and r["gpu_seconds"] == 0


def test_the_plan_counts_what_eight_lanes_add(built_lib):
    """hostsha_x8.cpp: a host thread with enough streams runs them eight at a time (a ctx plans with 200 % for files, 300 %
    for memory where the CPU has AVX-512; the exported call takes it as host_lane_gain_pct): a link-bound tree gives more
    of itself to the host, config 3 is modelled at what 16 threads then do, and a stream that dominates its thread's share
    still counts at one stream's rate."""
    lens = [MiB] * 10001
    _, r1 = _plan(lens, from_files=1)
    _, r2 = _plan(lens, from_files=1, host_lane_gain_pct=200)
    assert r2["host_bytes"] > 1.3 * r1["host_bytes"] and max(r2["gpu_seconds"], r2["host_seconds"]) < 0.95 * max(r1["gpu_seconds"], r1["host_seconds"])
    _, c3 = _plan([1 << 30] * 100, host_lane_gain_pct=300)
    assert c3["host_streams"] == 100 and c3["host_threads"] == 16 and 1.5 < c3["host_seconds"] < 2.2  # 7 streams a thread at 4.2 GB/s
    # the archive beside its tree: one stream, one core, one stream's rate -- whatever the gain
    _, pk = _plan([512 * MiB] + [MiB] * 512, from_files=1, host_lane_gain_pct=200)
    assert pk["host_streams"] == 1 and abs(pk["host_seconds"] - 512 * MiB / 1.4e9) < 0.02


def test_plan_is_deterministic_and_covers_every_stream(built_lib):
    import random
    rng = random.Random(5)
    lens = [rng.choice([0, 1, 127, 128, 4096, 70000, MiB, 9 * MiB]) for _ in range(3000)]
    a, ra = _plan(lens)
    b, rb = _plan(lens)
    assert a == b and ra == rb and len(a) == len(lens)
    assert ra["host_bytes"] == sum(l for l, h in zip(lens, a) if h) and ra["host_streams"] == sum(a)
    # longest first: no stream on the GPU is longer than a stream on the host
    gpu = [l for l, h in zip(lens, a) if not h]
    host = [l for l, h in zip(lens, a) if h]
    if gpu and host:
        assert max(gpu) <= min(host)
    assert _plan([], cpus=4)[0] == []


def test_cgroup_cpu_quota_on_fake_trees(built_lib, tmp_path):
    """The GPU box hands a 1-GPU job all 256 CPUs in its affinity mask and a CFS quota of 16 (profiles/r04_box_probe.txt):
    the planner must count 16."""
    from snappy_amd import _lib
    L = _lib.lib()
    # cgroup v2, quota on an ancestor
    root = tmp_path / "v2"
    (root / "job" / "step").mkdir(parents=True)
    (root / "cpu.max").write_text("max 100000\n")
    (root / "job" / "cpu.max").write_text("1600000 100000\n")
    (root / "job" / "step" / "cpu.max").write_text("max 100000\n")
    proc = tmp_path / "cg2"
    proc.write_text("0::/job/step\n")
    assert L.snaphash_cgroup_cpu_quota(str(root).encode(), str(proc).encode()) == 16
    (root / "job" / "step" / "cpu.max").write_text("250000 100000\n")  # the tightest binds; 2.5 CPUs -> 3
    assert L.snaphash_cgroup_cpu_quota(str(root).encode(), str(proc).encode()) == 3
    # cgroup v1
    root1 = tmp_path / "v1"
    (root1 / "cpu,cpuacct" / "docker" / "abc").mkdir(parents=True)
    (root1 / "cpu,cpuacct" / "docker" / "abc" / "cpu.cfs_quota_us").write_text("800000\n")
    (root1 / "cpu,cpuacct" / "docker" / "abc" / "cpu.cfs_period_us").write_text("100000\n")
    (root1 / "cpu,cpuacct" / "cpu.cfs_quota_us").write_text("-1\n")
    (root1 / "cpu,cpuacct" / "cpu.cfs_period_us").write_text("100000\n")
    proc1 = tmp_path / "cg1"
    proc1.write_text("12:cpuset:/docker/abc\n4:cpu,cpuacct:/docker/abc\n0::/\n")
    assert L.snaphash_cgroup_cpu_quota(str(root1).encode(), str(proc1).encode()) == 8
    # none
    proc0 = tmp_path / "cg0"
    proc0.write_text("0::/\n")
    assert L.snaphash_cgroup_cpu_quota(str(tmp_path / "nowhere").encode(), str(proc0).encode()) == 0
    assert 1 <= L.snaphash_usable_cpus() <= len(os.sched_getaffinity(0))


def test_cli_plan_verb_needs_no_device(built_lib, tmp_path):
    """`snaphash plan FILE...` (snappy_amd/cli/snaphash_cli.c): the plan of a call from plain C, over snaphash_plan_streams --
    it touches no device, so it runs here."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "snappy_amd", "bin", "snaphash")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "snappy_amd", "csrc")])
    small, big = tmp_path / "small.bin", tmp_path / "big.bin"
    small.write_bytes(b"x" * 3000)
    big.write_bytes(b"y" * (24 << 20))
    r = subprocess.run([exe, "-t", "4", "plan", str(small), str(big)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    lines = r.stdout.decode().splitlines()
    assert lines[1].startswith("host") and lines[1].endswith("big.bin")   # 24 MiB alone on the GPU: 0.57 s; on a host core 18 ms
    assert lines[0].split()[0] in ("gpu", "host") and b"modelled" in r.stderr


# ---- round 5: the model's constants are measured on the box, and a wrong one is corrected (planner.h PlanCalib) -------
def _calib():
    import ctypes
    from snappy_amd import _lib
    return _lib.PlanCalib(ctypes.sizeof(_lib.PlanCalib))


def _observe(c, what, nbytes, seconds):
    import ctypes
    from snappy_amd import _lib
    return _lib.lib().snaphash_calib_observe(ctypes.byref(c), what, float(nbytes), float(seconds))


def _apply(c, from_files, **kw):
    import ctypes
    from snappy_amd import _lib
    pm = _lib.PlanModel(ctypes.sizeof(_lib.PlanModel))
    pm.from_files = from_files
    for k, v in kw.items():
        setattr(pm, k, v)
    assert _lib.lib().snaphash_calib_apply(ctypes.byref(c), ctypes.byref(pm)) == 0
    return pm.gpu_link, pm.fill_rate


def test_a_wrong_link_constant_is_corrected_by_what_the_calls_measure(built_lib):
    """VERDICT r4 item 4.  The model's 54/55 GB/s is one lease's PCIe link.  On a box whose copies run at 25 GB/s the
    observations (HIP-event time of the H2D copies, one per staged call) move the calibrated link there -- the first one
    at once, outliers only a quarter of the way -- and the plan of config 2 changes with it: a slower link leaves more of
    the tree to the host threads."""
    c = _calib()
    assert _apply(c, 0) == (0.0, 0.0)                              # nothing measured: the model keeps its defaults
    assert _observe(c, 0, 1 << 20, 1e-5) == 0 and c.n_dma == 0     # too small to mean anything
    assert _observe(c, 0, 1 << 30, 1e-9) == 0 and c.n_dma == 0     # no box does that
    assert _observe(c, 0, 256 * MiB, 256 * MiB / 25e9) == 1
    link, _ = _apply(c, 0)
    assert abs(link - 25e9 * 55.0 / 56.7) < 1e6                    # the first observation stands
    assert _observe(c, 0, 256 * MiB, 256 * MiB / 50e9) == 1        # one odd call does not re-plan the box ...
    assert abs(c.dma - (0.75 * 25e9 + 0.25 * 50e9)) < 1e6
    for _ in range(12):                                            # ... a box that really is faster gets there
        _observe(c, 0, 256 * MiB, 256 * MiB / 50e9)
    assert 48e9 < c.dma <= 50e9
    # an explicit link in the model is the caller's: calibration fills only what is unset
    assert _apply(c, 0, gpu_link=40e9)[0] == 40e9

    lens = [MiB] * 10001
    slow = _calib()
    _observe(slow, 0, 256 * MiB, 256 * MiB / 25e9)
    link_slow, _ = _apply(slow, 1)
    _, r_default = _plan(lens, from_files=1, host_lane_gain_pct=240)
    _, r_slow = _plan(lens, from_files=1, host_lane_gain_pct=240, gpu_link=link_slow)
    assert r_slow["host_bytes"] > r_default["host_bytes"] * 1.5, (r_default, r_slow)
    # and the prediction follows: the same split costs the GPU part twice as long on the slow link
    _, g_fast = _plan(lens, from_files=1, host_threads=1, cpus=2)
    _, g_slow = _plan(lens, from_files=1, host_threads=1, cpus=2, gpu_link=link_slow)
    assert g_slow["gpu_seconds"] > 1.8 * g_fast["gpu_seconds"]


def test_fill_rate_calibration_and_what_it_changes(built_lib):
    """One fill thread's rate is measured inside the pipeline (wall x threads of a staged call's fills) and believed DOWN to
    half of the model's default, never up: the default is what makes the model's "cores" bound right beside host threads
    that compete with the fill, not what a thread can move; and an absurd observation (the first read of freshly
    written tmpfs files: 0.22 GB/s) must not send every later call to the host threads for good."""
    c = _calib()
    assert _apply(c, 0) == (0.0, 0.0) and _apply(c, 1) == (0.0, 0.0)
    assert _observe(c, 1, 64 * MiB, 64 * MiB / 13e9) == 1
    assert abs(_apply(c, 0)[1] - 9e9) < 1e6                        # a faster box leaves the default alone
    assert _apply(c, 1)[1] == 0.0                                  # no file has been read: files keep their default
    c = _calib()
    assert _observe(c, 1, 64 * MiB, 64 * MiB / 6e9) == 1
    assert abs(_apply(c, 0)[1] - 6e9) < 1e6                        # a slower box scales it down
    assert _observe(c, 2, 512 * MiB, 512 * MiB / 0.22e9) == 1
    assert abs(_apply(c, 1)[1] - 3.25e9) < 1e6 and abs(c.fill_files - 3.25e9) < 1e6   # ... but never below half of the default
    assert _observe(c, 2, 512 * MiB, 512 * MiB / 5.2e9) == 1       # and the next sane call is already a quarter of the way back
    assert abs(c.fill_files - (0.75 * 3.25e9 + 0.25 * 5.2e9)) < 1e6
    c.fill_files = 3.25e9
    assert _observe(c, 2, 100, 1.0) == 0 and _observe(c, 7, 1e9, 1.0) < 0
    # the link: a generation or two away from the 56.7 GB/s of the defaults, no further
    k = _calib()
    assert _observe(k, 0, 256 * MiB, 256 * MiB / 3e9) == 1
    assert abs(_apply(k, 0)[0] - 0.25 * 56.7e9 * 55.0 / 56.7) < 1e6 and abs(k.dma - 0.25 * 56.7e9) < 1e6   # cut where it enters
    # a fill thread at half of the model's rate makes the fill, not the link, the bound of the GPU part
    lens = [MiB] * 5000
    _, fast = _plan(lens, from_files=1, host_threads=1, cpus=2, fill_threads=4)
    _, slow = _plan(lens, from_files=1, host_threads=1, cpus=2, fill_threads=4, fill_rate=1.3e9)
    assert slow["gpu_seconds"] > 2.0 * fast["gpu_seconds"]
    # With the clamped worst case sixteen cores take a tree of 4 097 files whole -- and would never measure a fill again:
    # every such call moves the estimate a quarter of the way back, until the GPU part is tried (and measured) once more
    on_host, r = _plan([MiB] * 4097, from_files=1, host_lane_gain_pct=240, fill_rate=3.25e9)
    assert sum(on_host) == 4097 and r["gpu_seconds"] == 0
    for calls in range(1, 12):
        assert _observe(c, 4, 0, 0) == 1
        rate = _apply(c, 1)[1]
        on_host, r = _plan([MiB] * 4097, from_files=1, host_lane_gain_pct=240, fill_rate=rate)
        if sum(on_host) < 4097:
            break
    assert calls <= 6 and r["gpu_seconds"] > 0 and rate < 6.5e9, (calls, rate)
    assert _observe(c, 3, 0, 0) == 1 and abs(_apply(c, 0)[1] - (0.75 * 6e9 + 0.25 * 9e9)) < 1e6


def test_what_a_call_of_small_files_may_say_about_the_box(built_lib):
    """Round 5, found on the GPU box (profiles/r05_small_files.txt): after a tree of 5 000 x 8 KiB the ctx's link estimate was
    41 GB/s (three copies of 12 MiB measure their latency) and its fill rate at the floor (the threads had spent their time in
    open(), not in bytes) -- and the next big tree was planned with both.  A call is now sorted by what it can speak about:
    the link from copies of 32 MiB and more, the fill rate from streams of 256 KiB and more (net of the per-file cost), and a
    call of small files speaks about what a FILE costs a fill thread, which is calibrated too (both ways, 0.3 .. 2 x 10 us)."""
    import ctypes
    from snappy_amd import _lib
    L = _lib.lib()

    def call(c, files, nbytes, streams, copies, h2d_s, fill_s):
        assert L.snaphash_calib_observe_call(ctypes.byref(c), files, float(nbytes), float(streams), float(copies), float(h2d_s), float(fill_s)) == 0

    c = _calib()
    # 5 000 x 8 KiB: 41 MB in six copies at 30 GB/s (latency), 37 ms of fill threads: neither the link nor the fill RATE hears of it
    call(c, 1, 5000 * 8192, 5000, 6, 5000 * 8192 / 30e9, 0.037)
    assert c.n_dma == 0 and c.n_fill_files == 0 and c.dma == 0 and c.fill_files == 0
    # ... the per-file cost does: (37 ms - 41 MB / 6.5 GB/s) / 5 000 = 6.1 us
    assert c.n_fill_per_file == 1 and abs(c.fill_per_file - (0.037 - 5000 * 8192 / 6.5e9) / 5000) < 1e-9
    pm = _lib.PlanModel(ctypes.sizeof(_lib.PlanModel))
    pm.from_files = 1
    assert L.snaphash_calib_apply(ctypes.byref(c), ctypes.byref(pm)) == 0 and abs(pm.fill_per_file - c.fill_per_file) < 1e-12
    pm = _lib.PlanModel(ctypes.sizeof(_lib.PlanModel))
    assert L.snaphash_calib_apply(ctypes.byref(c), ctypes.byref(pm)) == 0 and pm.fill_per_file == 0.0   # memory sources open nothing
    # config 2 (10 001 x 1 MiB in 42 copies): all three constants, the fill rate net of the files' own cost
    call(c, 1, 10001 * MiB, 10001, 42, 10001 * MiB / 50e9, 10001 * MiB / 5e9 + 10001 * c.fill_per_file)
    assert c.n_dma == 1 and abs(c.dma - 50e9) < 1e6 and c.n_fill_files == 1 and abs(c.fill_files - 5e9) < 1e7 and c.n_fill_per_file == 1
    # a few dozen files say nothing about a file's cost; 64 KiB .. 256 KiB on average speaks about neither
    before = (c.fill_per_file, c.fill_files)
    call(c, 1, 100 * 4096, 100, 1, 1e-4, 0.01)
    call(c, 1, 5000 * (128 << 10), 5000, 30, 0.02, 0.5)
    assert (c.fill_per_file, c.fill_files) == before and c.n_dma == 1
    # an absurd one (a cold cache: 1 ms a file) is cut where it enters, and moves the estimate a quarter of the way
    call(c, 1, 5000 * 8192, 5000, 6, 1e-3, 5.0)
    assert abs(c.fill_per_file - (0.75 * before[0] + 0.25 * 20e-6)) < 1e-9
    assert _observe(c, 6, 10, 1.0) == 0 and _observe(c, 6, 5000, 5000 * 1e-9) == 0   # too few files; no box opens a file in a nanosecond
    # what it changes: the GPU part of 20 000 small files is modelled by its opens
    lens = [8192] * 20000
    _, slow = _plan(lens, from_files=1, host_threads=1, cpus=2, fill_threads=12)
    _, fast = _plan(lens, from_files=1, host_threads=1, cpus=2, fill_threads=12, fill_per_file=5e-6)
    assert fast["gpu_seconds"] < 0.7 * slow["gpu_seconds"]


def test_abi4_sized_structs_are_still_taken(built_lib):
    """A caller built against ABI 4 passes the shorter snaphash_plan_model (no fill_rate)."""
    import ctypes
    from snappy_amd import _lib
    pm = _lib.PlanModel()
    pm.struct_size = _lib.PlanModel.fill_rate.offset
    pm.cpus = 16
    arr = (ctypes.c_uint64 * 3)(MiB, MiB, 77)
    assert _lib.lib().snaphash_plan_streams(arr, 3, ctypes.byref(pm), None) == 0
    pm.struct_size = 8
    assert _lib.lib().snaphash_plan_streams(arr, 3, ctypes.byref(pm), None) == _lib.EINVAL


def test_host_rate_correction_from_what_the_host_part_took(built_lib):
    """The lane gain of the eight-stream host hasher is one box's number: a host part planned at 147 ms that took 130 says the
    threads of this box are 13 % faster than modelled, and the next plan gives them more (the model's
    prediction is set beside what happened, and learns from it).  Bounded (0.6 .. 1.6), parts under 5 ms say nothing."""
    c = _calib()
    assert _observe(c, 5, 0.002, 0.001) == 0 and c.n_host == 0
    assert _observe(c, 5, 0.147, 0.130) == 1 and abs(c.host_gain - 147.0 / 130.0) < 1e-9
    assert _observe(c, 5, 0.130, 0.130) == 1 and abs(c.host_gain - 147.0 / 130.0) < 1e-9   # a plan that came true changes nothing
    assert _observe(c, 5, 0.100, 1.000) == 1 and c.host_gain >= 0.6                        # one awful call: cut, and a quarter of the way
    k = _calib()
    for _ in range(20):
        _observe(k, 5, 0.100, 0.020)
    assert abs(k.host_gain - 1.6) < 1e-9                                                   # never beyond the bound
    # and the plan of config 2 follows: faster host threads take more of the tree
    import ctypes
    from snappy_amd import _lib
    lens = [MiB] * 10001
    _, base = _plan(lens, from_files=1, host_lane_gain_pct=240, host_rate=1.26e9)
    _, more = _plan(lens, from_files=1, host_lane_gain_pct=240, host_rate=1.26e9 * 1.3)
    assert more["host_bytes"] > base["host_bytes"]


@pytest.mark.parametrize("n", [300, 4095, 4096, 20000])
def test_the_host_takes_the_longest_streams_and_ties_go_in_list_order(built_lib, n):
    """Round 5: lists of 4 096 streams and more are ordered by a radix sort (config 5's 100 000 streams took 14 ms to plan
    in front of a 190 ms call, 6 of them in a merge sort through an index).  Whatever orders the list, what moves to the
    host threads is a PREFIX of "longest first, equal lengths in list order" -- also with a handful of distinct lengths,
    where nearly every compare is a tie."""
    import numpy as np
    rng = np.random.default_rng(n)
    shapes = {
        "ties": rng.integers(1, 6, size=n) * (3 * MiB),                       # five distinct lengths
        "zipf": np.maximum(1024, (2**28 / rng.permutation(np.arange(1, n + 1))).astype(np.int64)),  # config 5's shape, shuffled
        "head": np.concatenate([[5 << 30], rng.integers(0, 1 << 16, size=n - 1)]),  # one stream of more than 32 bits of length
    }
    for name, lens in shapes.items():
        on_host, r = _plan([int(x) for x in lens], from_files=0, host_rate=1.4e9, host_lane_gain_pct=240)
        k = r["host_streams"]
        assert sum(on_host) == k and k > 0, name
        want = np.argsort(-lens.astype(np.float64), kind="stable")[:k]
        assert sorted(np.flatnonzero(np.array(on_host))) == sorted(want.tolist()), name


def test_a_dominant_stream_is_planned_as_before_by_the_shortened_search(built_lib):
    """Config 5 (100 000 files, a 255 MiB head): from the moment the head is on a host thread it alone is the host part's
    makespan, and once that is what the split waits for no further stream can shorten the call -- the search over thread
    counts stops there instead of trying all 100 000 streams for each (round 5: 23 -> 2 ms in this container, 14 of a
    203 ms call on the GPU box before).  The decision is what it was: the head and the few dozen streams the GPU would take
    longer for than the head takes a core go to host threads, and the host part is modelled at the head's time."""
    import numpy as np
    from snappy_amd import synthetic
    lens = [int(x) for x in synthetic.config_sizes("C5")[:-1]]
    on_host, r = _plan(lens, from_files=0, host_rate=1.4e9, host_lane_gain_pct=240, fill_threads=6)
    assert 20 <= r["host_streams"] <= 60 and on_host[int(np.argmax(lens))] == 1
    assert r["host_seconds"] < 1.05 * max(lens) / 1.4e9 + 1e-3  # the head alone
    assert r["gpu_seconds"] < 1.1 * r["host_seconds"]
