"""ctypes binding of libsnaphash.so (include/snaphash.h).

The library is the product: HIP kernels for gfx950 behind a C ABI.  There is
no Python or CPU fallback -- if the shared object is missing this module
raises at import of the symbol table, and if no MI355X is visible
``Context()`` raises ``SnaphashError(EDEVICE)``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNAPHASH_LIB: developer override to A/B another build of the same ABI (tools/ab_bench.sh)
LIB_PATH = os.environ.get("SNAPHASH_LIB") or os.path.join(_HERE, "libsnaphash.so")

OK, EINVAL, ENOMEM, EIO, EDEVICE, EMODE, ENAME, EPARSE, EMISMATCH = 0, -1, -2, -3, -4, -5, -6, -7, -8
KERNEL_AUTO, KERNEL_WIDE, KERNEL_SPLIT, KERNEL_PAIR, KERNEL_QUAD = 0, 1, 2, 3, 4
KERNEL_NAMES = {KERNEL_WIDE: "sha512_wide_kernel", KERNEL_SPLIT: "sha512_split_kernel<false>",
                KERNEL_PAIR: "sha512_split_kernel<true>", KERNEL_QUAD: "sha512_quad_kernel"}

# every symbol include/snaphash.h declares
EXPORTS = [
    "snaphash_init", "snaphash_destroy", "snaphash_abi_version",
    "snaphash_sha512_files", "snaphash_sha512_buffers", "snaphash_sha512_device", "snaphash_sync",
    "snaphash_tree", "snaphash_write_hashes", "snaphash_verify", "snaphash_free",
    "snaphash_files_equal", "snaphash_ranges_equal_device", "snaphash_dir_updated",
    "snaphash_walk", "snaphash_records_count", "snaphash_records_get", "snaphash_records_free",
    "snaphash_emit_yaml", "snaphash_parse_yaml", "snaphash_records_sha512_hex", "snaphash_mode_string", "snaphash_mode_parse", "snaphash_lpt_assign",
    "snaphash_fill_synthetic_device", "snaphash_strerror", "snaphash_last_error", "snaphash_get_stats",
    "snaphash_get_stats_ex", "snaphash_get_device_stats", "snaphash_tree_ex",
    "snaphash_batch_begin", "snaphash_batch_append", "snaphash_batch_end", "snaphash_batch_finish", "snaphash_batch_abort",
    "snaphash_tar_create", "snaphash_tar_create_fn", "snaphash_gzip_buffer", "snaphash_get_targz_stats",
    "snaphash_get_engine_info", "snaphash_numa_probe",
    "snaphash_get_engine_cpus", "snaphash_numa_slice", "snaphash_plan_streams", "snaphash_usable_cpus", "snaphash_cgroup_cpu_quota",
    "snaphash_shard_plan", "snaphash_shard_rows", "snaphash_shard_count", "snaphash_shard_streams", "snaphash_shard_bytes",
    "snaphash_shard_path", "snaphash_shard_hash", "snaphash_shard_emit", "snaphash_shard_free",
    # ABI 5
    "snaphash_shard_set_local_ranks", "snaphash_shard_fingerprint", "snaphash_get_plan_model", "snaphash_calib_observe_call",
    "snaphash_shard_list", "snaphash_shard_plan_from",
    "snaphash_calib_observe", "snaphash_calib_apply", "snaphash_get_calib",
]
FLAG_CHECK_GATHER, FLAG_NO_RCCL, FLAG_FORCE_GATHER, FLAG_GPU_ONLY, FLAG_NO_NUMA, FLAG_KEEP_RLIMIT = 1, 2, 4, 8, 16, 32


class Config(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("staging_bytes", ctypes.c_uint64),
                ("kernel", ctypes.c_uint32), ("deflate_depth", ctypes.c_uint32), ("stream", ctypes.c_void_p),
                # ABI 2
                ("devices", ctypes.POINTER(ctypes.c_int32)), ("n_devices", ctypes.c_uint32),
                ("host_threads", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("reserved2", ctypes.c_uint32)]


class StatsEx(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("n_devices", ctypes.c_uint32), ("gather_kind", ctypes.c_uint32),
                ("gather_checked", ctypes.c_uint32), ("gather_ms", ctypes.c_double), ("gpu_bytes", ctypes.c_uint64),
                ("host_bytes", ctypes.c_uint64), ("host_streams", ctypes.c_uint64), ("reserved3", ctypes.c_uint64),
                ("host_ms", ctypes.c_double),
                # ABI 5: the plan's prediction beside what the call took
                ("planned_gpu_ms", ctypes.c_double), ("planned_host_ms", ctypes.c_double), ("planned_threads", ctypes.c_uint32),
                ("host_threads_run", ctypes.c_uint32), ("gpu_ms", ctypes.c_double), ("hash_ms", ctypes.c_double),
                ("plan_ms", ctypes.c_double)]


class EngineInfo(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("numa_node", ctypes.c_int32),
                ("staging_node", ctypes.c_int32), ("fill_threads", ctypes.c_uint32), ("n_cpus", ctypes.c_uint32),
                ("pci_bus_id", ctypes.c_char * 32), ("pinned_bytes", ctypes.c_uint64), ("hbm_bytes", ctypes.c_uint64)]


class PlanModel(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("n_devices", ctypes.c_uint32), ("cpus", ctypes.c_uint32),
                ("fill_threads", ctypes.c_uint32), ("host_threads", ctypes.c_uint32), ("from_files", ctypes.c_uint32),
                ("host_rate", ctypes.c_double), ("gpu_stream_rate", ctypes.c_double), ("gpu_link", ctypes.c_double),
                ("gpu_latency", ctypes.c_double),
                ("gpu_seconds", ctypes.c_double), ("host_seconds", ctypes.c_double), ("host_streams", ctypes.c_uint64),
                ("host_bytes", ctypes.c_uint64), ("host_threads_used", ctypes.c_uint32), ("host_lane_gain_pct", ctypes.c_uint32),
                ("fill_rate", ctypes.c_double), ("fill_per_file", ctypes.c_double)]  # ABI 5


class PlanCalib(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("n_dma", ctypes.c_uint32), ("n_fill_mem", ctypes.c_uint32),
                ("n_fill_files", ctypes.c_uint32), ("dma", ctypes.c_double), ("fill_mem", ctypes.c_double),
                ("fill_files", ctypes.c_double), ("host_gain", ctypes.c_double), ("n_host", ctypes.c_uint32), ("n_fill_per_file", ctypes.c_uint32),
                ("fill_per_file", ctypes.c_double)]


class Stats(ctypes.Structure):
    _fields_ = [("bytes_hashed", ctypes.c_uint64), ("blocks", ctypes.c_uint64), ("streams", ctypes.c_uint64),
                ("launches", ctypes.c_uint32), ("kernel_used", ctypes.c_uint32), ("kernel_ms", ctypes.c_double),
                ("h2d_ms", ctypes.c_double), ("wall_ms", ctypes.c_double)]


class TargzStats(ctypes.Structure):
    _fields_ = [("tar_bytes", ctypes.c_uint64), ("gz_bytes", ctypes.c_uint64), ("members", ctypes.c_uint64),
                ("chunks", ctypes.c_uint64), ("stored_chunks", ctypes.c_uint64), ("deflate_ms", ctypes.c_double),
                ("fill_ms", ctypes.c_double), ("wall_ms", ctypes.c_double)]


class Mismatch(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("reserved", ctypes.c_int32), ("name", ctypes.c_char * 4096)]


class Record(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("st_mode", ctypes.c_uint32), ("is_regular", ctypes.c_int32),
                ("size", ctypes.c_int64), ("path", ctypes.c_char_p)]


class SnaphashError(Exception):
    def __init__(self, code, message=""):
        self.code = code
        super().__init__("snaphash error %d (%s)%s" % (code, strerror(code), (": " + message) if message else ""))


_lib = None


def lib():
    """Load libsnaphash.so; raises OSError loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: build it with `make -C snappy_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64, and a second runtime
    # loaded before it ends up without a device ("no ROCm-capable device is detected").  In the
    # Python mirror torch is always around (device memory, streams, torch.distributed), so load
    # it first and let libsnaphash.so bind to the runtime that is already in the process.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, u64p = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)
    L.snaphash_init.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    L.snaphash_destroy.argtypes = [vp]
    L.snaphash_destroy.restype = None
    L.snaphash_abi_version.argtypes = []
    L.snaphash_sha512_files.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), sz, vp, ctypes.POINTER(ctypes.c_int32)]
    L.snaphash_sha512_buffers.argtypes = [vp, ctypes.POINTER(vp), u64p, sz, vp]
    L.snaphash_sha512_device.argtypes = [vp, vp, vp, vp, sz, vp]
    L.snaphash_sync.argtypes = [vp]
    L.snaphash_tree.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.snaphash_write_hashes.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    L.snaphash_verify.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, sz, ctypes.POINTER(Mismatch)]
    L.snaphash_files_equal.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p), sz, vp]
    L.snaphash_ranges_equal_device.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp]
    L.snaphash_dir_updated.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(vp),
                                       ctypes.POINTER(sz)]
    L.snaphash_free.argtypes = [vp]
    L.snaphash_free.restype = None
    L.snaphash_walk.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.snaphash_records_count.argtypes = [vp]
    L.snaphash_records_count.restype = sz
    L.snaphash_records_get.argtypes = [vp, sz, ctypes.POINTER(Record)]
    L.snaphash_records_free.argtypes = [vp]
    L.snaphash_records_free.restype = None
    L.snaphash_emit_yaml.argtypes = [vp, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.snaphash_parse_yaml.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(vp), ctypes.c_char_p]
    L.snaphash_records_sha512_hex.argtypes = [vp, sz]
    L.snaphash_records_sha512_hex.restype = ctypes.c_char_p
    L.snaphash_mode_string.argtypes = [ctypes.c_uint32, ctypes.c_char_p]
    L.snaphash_mode_parse.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32)]
    L.snaphash_lpt_assign.argtypes = [vp, sz, ctypes.c_int, vp]
    L.snaphash_fill_synthetic_device.argtypes = [vp, vp, vp, vp, vp, sz]
    L.snaphash_strerror.argtypes = [ctypes.c_int]
    L.snaphash_strerror.restype = ctypes.c_char_p
    L.snaphash_last_error.argtypes = [vp]
    L.snaphash_last_error.restype = ctypes.c_char_p
    L.snaphash_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.snaphash_get_stats.restype = None
    L.snaphash_get_stats_ex.argtypes = [vp, ctypes.POINTER(StatsEx)]
    L.snaphash_get_device_stats.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(Stats)]
    L.snaphash_tree_ex.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int,
                                   ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.snaphash_batch_begin.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.snaphash_batch_append.argtypes = [vp, sz, vp, sz]
    L.snaphash_batch_end.argtypes = [vp, sz]
    L.snaphash_batch_finish.argtypes = [vp, vp]
    L.snaphash_batch_abort.argtypes = [vp]
    L.snaphash_batch_abort.restype = None
    L.snaphash_tar_create.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(vp),
                                      ctypes.POINTER(sz), ctypes.c_char_p]
    L.snaphash_gzip_buffer.argtypes = [vp, vp, sz, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.snaphash_get_targz_stats.argtypes = [vp, ctypes.POINTER(TargzStats)]
    L.snaphash_get_targz_stats.restype = None
    L.snaphash_get_engine_info.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(EngineInfo)]
    L.snaphash_numa_probe.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32), vp, sz, ctypes.POINTER(sz)]
    L.snaphash_shard_plan.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp)]
    for f in ("rows", "count", "streams"):
        getattr(L, "snaphash_shard_" + f).argtypes = [vp]
        getattr(L, "snaphash_shard_" + f).restype = sz
    L.snaphash_shard_bytes.argtypes = [vp]
    L.snaphash_shard_bytes.restype = ctypes.c_uint64
    L.snaphash_shard_path.argtypes = [vp, sz]
    L.snaphash_shard_path.restype = ctypes.c_char_p
    L.snaphash_shard_hash.argtypes = [vp, vp, vp]
    L.snaphash_shard_emit.argtypes = [vp, vp, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.snaphash_shard_free.argtypes = [vp]
    L.snaphash_shard_free.restype = None
    L.snaphash_numa_slice.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint32, vp, sz, ctypes.POINTER(sz)]
    L.snaphash_get_engine_cpus.argtypes = [vp, ctypes.c_uint32, vp, sz, ctypes.POINTER(sz)]
    L.snaphash_plan_streams.argtypes = [u64p, sz, ctypes.POINTER(PlanModel), vp]
    L.snaphash_shard_list.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.snaphash_shard_plan_from.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp), ctypes.POINTER(sz),
                                           ctypes.POINTER(vp)]
    L.snaphash_shard_set_local_ranks.argtypes = [vp, ctypes.c_uint32]
    L.snaphash_shard_fingerprint.argtypes = [vp]
    L.snaphash_shard_fingerprint.restype = ctypes.c_uint64
    L.snaphash_get_plan_model.argtypes = [vp, ctypes.c_int, ctypes.POINTER(PlanModel)]
    L.snaphash_calib_observe.argtypes = [ctypes.POINTER(PlanCalib), ctypes.c_int, ctypes.c_double, ctypes.c_double]
    L.snaphash_calib_observe_call.argtypes = [ctypes.POINTER(PlanCalib), ctypes.c_int] + [ctypes.c_double] * 5
    L.snaphash_calib_apply.argtypes = [ctypes.POINTER(PlanCalib), ctypes.POINTER(PlanModel)]
    L.snaphash_get_calib.argtypes = [vp, ctypes.POINTER(PlanCalib)]
    L.snaphash_usable_cpus.argtypes = []
    L.snaphash_usable_cpus.restype = ctypes.c_uint32
    L.snaphash_cgroup_cpu_quota.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.snaphash_cgroup_cpu_quota.restype = ctypes.c_uint32
    _lib = L
    return L


def plan_streams(lens, **model):
    """snaphash_plan_streams: what the planner decides for streams of these lengths (host-only, no device needed).
    Returns (on_host list of 0/1, dict of the model's outputs)."""
    n = len(lens)
    arr = (ctypes.c_uint64 * max(n, 1))(*[int(x) for x in lens])
    pm = PlanModel()
    pm.struct_size = ctypes.sizeof(PlanModel)
    for k, v in model.items():
        setattr(pm, k, v)
    flags = ctypes.create_string_buffer(max(n, 1))
    rc = lib().snaphash_plan_streams(arr, n, ctypes.byref(pm), flags)
    if rc:
        raise SnaphashError(rc)
    return list(flags.raw[:n]), {"gpu_seconds": pm.gpu_seconds, "host_seconds": pm.host_seconds, "host_streams": pm.host_streams,
                                 "host_bytes": pm.host_bytes, "host_threads": pm.host_threads_used}


def strerror(code):
    try:
        return lib().snaphash_strerror(code).decode()
    except OSError:
        return "?"


class Context:
    """One snaphash_ctx: one GPU (device=) or several (devices=[...], [-1] = all visible; the file list of a
    call is then LPT-sharded inside the library and the digests gathered over RCCL), one call in flight.
    host_threads: 0 = the library's default (a stream that would set the makespan of its batch all by itself -- the
    archive next to its tree -- is hashed on a host thread), N > 0 = N threads and the full planner;
    flags=FLAG_GPU_ONLY keeps every byte on the GPU (kernel parity tests, roofline runs)."""

    # flags a Context gets when the caller names none: 0 = the library's defaults.  The -m gpu suite sets it to
    # FLAG_GPU_ONLY (tests/conftest.py) so that every test exercises the HIP kernels unless it asks for the default.
    DEFAULT_FLAGS = 0

    def __init__(self, device=-1, staging_bytes=0, kernel=KERNEL_AUTO, stream=None, devices=None, host_threads=0, flags=None, deflate_depth=0):
        if flags is None:
            flags = Context.DEFAULT_FLAGS
        cfg = Config(ctypes.sizeof(Config), device, staging_bytes, kernel, deflate_depth, stream)
        if devices is not None:
            self._devs = (ctypes.c_int32 * len(devices))(*devices)
            cfg.devices = ctypes.cast(self._devs, ctypes.POINTER(ctypes.c_int32))
            cfg.n_devices = len(devices)
        cfg.host_threads = host_threads
        cfg.flags = flags
        h = ctypes.c_void_p()
        rc = lib().snaphash_init(ctypes.byref(cfg), ctypes.byref(h))
        if rc:
            raise SnaphashError(rc, "snaphash_init: " + lib().snaphash_last_error(None).decode(errors="replace"))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().snaphash_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc:
            raise SnaphashError(rc, lib().snaphash_last_error(self._h).decode(errors="replace"))

    # ---- primitive -----------------------------------------------------------------
    def sha512_files(self, paths):
        """-> list of 64-byte digests; raises OSError(errno) for the first unreadable file."""
        n = len(paths)
        arr = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(p) for p in paths])
        out = ctypes.create_string_buffer(64 * max(n, 1))
        status = (ctypes.c_int32 * max(n, 1))()
        rc = lib().snaphash_sha512_files(self._h, arr, n, out, status)
        if rc == EIO:
            for i in range(n):
                if status[i]:
                    raise OSError(status[i], os.strerror(status[i]), os.fsdecode(paths[i]))
        self._check(rc)
        return [out.raw[64 * i:64 * i + 64] for i in range(n)]

    def sha512_buffers(self, bufs):
        n = len(bufs)
        keep = [bytes(b) if not isinstance(b, (bytes, bytearray)) else b for b in bufs]
        cbufs = [(ctypes.c_char * max(len(b), 1)).from_buffer_copy(b if len(b) else b"\0") for b in keep]
        ptrs = (ctypes.c_void_p * max(n, 1))(*[ctypes.addressof(c) for c in cbufs])
        lens = (ctypes.c_uint64 * max(n, 1))(*[len(b) for b in keep])
        out = ctypes.create_string_buffer(64 * max(n, 1))
        self._check(lib().snaphash_sha512_buffers(self._h, ptrs, lens, n, out))
        return [out.raw[64 * i:64 * i + 64] for i in range(n)]

    def sha512_device(self, d_base, offsets, lens, d_digests):
        """HBM-resident batch.  d_base / d_digests: device addresses (int);
        offsets / lens: contiguous numpy uint64 arrays.  Asynchronous: call sync()."""
        n = len(offsets)
        self._check(lib().snaphash_sha512_device(self._h, d_base, offsets.ctypes.data, lens.ctypes.data, n, d_digests))

    def sync(self):
        self._check(lib().snaphash_sync(self._h))

    def fill_synthetic_device(self, d_base, offsets, lens, file_index):
        self._check(lib().snaphash_fill_synthetic_device(self._h, d_base, offsets.ctypes.data, lens.ctypes.data,
                                                         file_index.ctypes.data, len(offsets)))

    # ---- pass ------------------------------------------------------------------------
    def tree(self, build_dir, data_tar):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(lib().snaphash_tree(self._h, os.fsencode(build_dir), os.fsencode(data_tar), ctypes.byref(p),
                                        ctypes.byref(n)))
        try:
            return ctypes.string_at(p.value, n.value)
        finally:
            lib().snaphash_free(p)

    def write_hashes(self, build_dir, data_tar):
        self._check(lib().snaphash_write_hashes(self._h, os.fsencode(build_dir), os.fsencode(data_tar)))

    def tree_ex(self, build_dir, data_tar=None, archive_digest=None, write=False):
        """snaphash_tree_ex: the archive digest may be supplied by the caller (data_tar=None)."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(lib().snaphash_tree_ex(self._h, os.fsencode(build_dir), os.fsencode(data_tar) if data_tar else None,
                                           archive_digest, 1 if write else 0, ctypes.byref(p), ctypes.byref(n)))
        try:
            return ctypes.string_at(p.value, n.value)
        finally:
            lib().snaphash_free(p)

    def gzip_buffer(self, data):
        """One gzip member of `data`, DEFLATE on the GPU (row f3)."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        buf = (ctypes.c_char * max(len(data), 1)).from_buffer_copy(data if len(data) else b"\0")
        self._check(lib().snaphash_gzip_buffer(self._h, ctypes.addressof(buf), len(data), ctypes.byref(p), ctypes.byref(n)))
        try:
            return ctypes.string_at(p.value, n.value)
        finally:
            lib().snaphash_free(p)

    def tar_create(self, tarname, source_dir, exclude_prefix=None, with_hashes=False):
        """tarCreate (clickdeb/deb.go:261-344).  with_hashes: also hashes.yaml from the same read.
        -> (yaml bytes or None, archive digest (64 bytes))."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        dig = ctypes.create_string_buffer(64)
        self._check(lib().snaphash_tar_create(self._h, os.fsencode(tarname), os.fsencode(source_dir),
                                              os.fsencode(exclude_prefix) if exclude_prefix else None,
                                              ctypes.byref(p) if with_hashes else None, ctypes.byref(n), dig))
        try:
            return (ctypes.string_at(p.value, n.value) if with_hashes else None), dig.raw
        finally:
            if with_hashes:
                lib().snaphash_free(p)

    def tar_create_fn(self, tarname, source_dir, keep=None, with_hashes=False):
        """tarCreate with the reference's own exclude function: keep(path) -> bool (clickdeb/deb.go:261, 295-299)."""
        KEEP = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p)
        cb = KEEP((lambda path, _u: 1 if keep(os.fsdecode(path)) else 0) if keep else 0)
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        dig = ctypes.create_string_buffer(64)
        f = lib().snaphash_tar_create_fn
        f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, KEEP, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_char_p]
        self._check(f(self._h, os.fsencode(tarname), os.fsencode(source_dir), cb, None,
                      ctypes.cast(ctypes.byref(p), ctypes.c_void_p) if with_hashes else None,
                      ctypes.cast(ctypes.byref(n), ctypes.c_void_p), dig))
        try:
            return (ctypes.string_at(p.value, n.value) if with_hashes else None), dig.raw
        finally:
            if with_hashes:
                lib().snaphash_free(p)

    def targz_stats(self):
        s = TargzStats()
        lib().snaphash_get_targz_stats(self._h, ctypes.byref(s))
        return {f[0]: getattr(s, f[0]) for f in TargzStats._fields_}

    def batch(self, n_streams):
        """Streaming batch (row f2): feed chunks as another pass reads them."""
        return Batch(self, n_streams)

    def verify(self, inst_dir, yaml_bytes, data_tar=None):
        """-> None when the tree matches, else (kind, name) of the first mismatch."""
        m = Mismatch()
        rc = lib().snaphash_verify(self._h, os.fsencode(inst_dir), os.fsencode(data_tar) if data_tar else None,
                                   yaml_bytes, len(yaml_bytes), ctypes.byref(m))
        if rc == EMISMATCH:
            return (m.kind, m.name.decode(errors="replace"))
        self._check(rc)
        return None

    # ---- neighbouring scan: helpers.FilesAreEqual / DirUpdated ------------------------------
    def files_equal(self, pairs):
        """[(a, b)] -> [bool]; like helpers.FilesAreEqual any open/stat/read error means False."""
        n = len(pairs)
        a = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(p[0]) for p in pairs])
        b = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(p[1]) for p in pairs])
        out = ctypes.create_string_buffer(max(n, 1))
        self._check(lib().snaphash_files_equal(self._h, a, b, n, out))
        return [bool(x) for x in out.raw[:n]]

    def ranges_equal_device(self, d_a, off_a, d_b, off_b, lens, d_equal):
        self._check(lib().snaphash_ranges_equal_device(self._h, d_a, off_a.ctypes.data, d_b, off_b.ctypes.data,
                                                       lens.ctypes.data, len(lens), d_equal))

    def dir_updated(self, dir_a, dir_b, pfx=""):
        """helpers.DirUpdated -> {name: True} like the Go map."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(lib().snaphash_dir_updated(self._h, os.fsencode(dir_a), os.fsencode(dir_b), pfx.encode(),
                                               ctypes.byref(p), ctypes.byref(n)))
        try:
            names, addr = {}, p.value
            for _ in range(n.value):
                s_ = ctypes.string_at(addr)
                names[s_.decode(errors="surrogateescape")] = True
                addr += len(s_) + 1
            return names
        finally:
            lib().snaphash_free(p)

    def stats(self):
        s = Stats()
        lib().snaphash_get_stats(self._h, ctypes.byref(s))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_}

    def stats_ex(self):
        s = StatsEx(ctypes.sizeof(StatsEx))
        self._check(lib().snaphash_get_stats_ex(self._h, ctypes.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in StatsEx._fields_}

    def plan_model(self, from_files):
        """snaphash_get_plan_model: what this ctx plans a call with right now (calibrated link and fill rates included);
        its fields are keyword arguments of plan_streams()."""
        pm = PlanModel(ctypes.sizeof(PlanModel))
        self._check(lib().snaphash_get_plan_model(self._h, 1 if from_files else 0, ctypes.byref(pm)))
        return {k: getattr(pm, k) for k in ("n_devices", "cpus", "fill_threads", "host_threads", "from_files", "host_rate",
                                            "gpu_stream_rate", "gpu_link", "gpu_latency", "host_lane_gain_pct", "fill_rate", "fill_per_file")}

    def calib(self):
        c = PlanCalib(ctypes.sizeof(PlanCalib))
        self._check(lib().snaphash_get_calib(self._h, ctypes.byref(c)))
        return {f[0]: getattr(c, f[0]) for f in PlanCalib._fields_ if f[0] != "struct_size"}

    def engine_info(self, i):
        e = EngineInfo(ctypes.sizeof(EngineInfo))
        self._check(lib().snaphash_get_engine_info(self._h, i, ctypes.byref(e)))
        out = {f[0]: getattr(e, f[0]) for f in EngineInfo._fields_}
        out["pci_bus_id"] = e.pci_bus_id.decode()
        return out

    def engine_cpus(self, i):
        n = ctypes.c_size_t()
        self._check(lib().snaphash_get_engine_cpus(self._h, i, None, 0, ctypes.byref(n)))
        buf = (ctypes.c_int32 * max(n.value, 1))()
        self._check(lib().snaphash_get_engine_cpus(self._h, i, buf, n.value, ctypes.byref(n)))
        return list(buf[:n.value])

    def device_stats(self, i):
        s, d = Stats(), ctypes.c_int32()
        self._check(lib().snaphash_get_device_stats(self._h, i, ctypes.byref(d), ctypes.byref(s)))
        out = {f[0]: getattr(s, f[0]) for f in Stats._fields_}
        out["device"] = d.value
        return out


class Batch:
    """snaphash_batch_*: hash.Hash-style appends into many streams, digests at finish()."""

    def __init__(self, ctx, n_streams):
        self._ctx, self.n = ctx, n_streams
        h = ctypes.c_void_p()
        ctx._check(lib().snaphash_batch_begin(ctx._h, n_streams, ctypes.byref(h)))
        self._h = h

    def append(self, stream, data):
        buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if len(data) else None
        self._ctx._check(lib().snaphash_batch_append(self._h, stream, ctypes.addressof(buf) if buf is not None else None, len(data)))

    def append_ptr(self, stream, addr, n):
        self._ctx._check(lib().snaphash_batch_append(self._h, stream, addr, n))

    def end(self, stream):
        self._ctx._check(lib().snaphash_batch_end(self._h, stream))

    def finish(self):
        out = ctypes.create_string_buffer(64 * max(self.n, 1))
        h, self._h = self._h, None
        self._ctx._check(lib().snaphash_batch_finish(h, out))
        return [out.raw[64 * i:64 * i + 64] for i in range(self.n)]

    def abort(self):
        if self._h:
            lib().snaphash_batch_abort(self._h)
            self._h = None


# ---- host-only entry points (no device) ----------------------------------------------

def walk(build_dir):
    """filepath.Walk as writeHashes drives it -> list of dicts in walk order."""
    h = ctypes.c_void_p()
    rc = lib().snaphash_walk(os.fsencode(build_dir), ctypes.byref(h))
    if rc:
        raise SnaphashError(rc, build_dir)
    try:
        out = []
        r = Record()
        for i in range(lib().snaphash_records_count(h)):
            lib().snaphash_records_get(h, i, ctypes.byref(r))
            out.append({"name": r.name.decode(errors="surrogateescape"), "st_mode": r.st_mode,
                        "is_regular": bool(r.is_regular), "size": r.size,
                        "path": r.path.decode(errors="surrogateescape")})
        return out
    finally:
        lib().snaphash_records_free(h)


def emit_yaml(build_dir, archive_digest, file_digests):
    """yaml.Marshal(hashesYaml) for the tree at build_dir with the given raw digests."""
    h = ctypes.c_void_p()
    rc = lib().snaphash_walk(os.fsencode(build_dir), ctypes.byref(h))
    if rc:
        raise SnaphashError(rc, build_dir)
    try:
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        blob = b"".join(file_digests)
        rc = lib().snaphash_emit_yaml(h, archive_digest, blob if blob else None, ctypes.byref(p), ctypes.byref(n))
        if rc:
            raise SnaphashError(rc)
        try:
            return ctypes.string_at(p.value, n.value)
        finally:
            lib().snaphash_free(p)
    finally:
        lib().snaphash_records_free(h)


def parse_yaml(text):
    """yaml.Unmarshal into hashesYaml -> (archive_hex, [record dicts])."""
    h = ctypes.c_void_p()
    arch = ctypes.create_string_buffer(129)
    rc = lib().snaphash_parse_yaml(text, len(text), ctypes.byref(h), arch)
    if rc:
        raise SnaphashError(rc)
    try:
        out = []
        r = Record()
        for i in range(lib().snaphash_records_count(h)):
            lib().snaphash_records_get(h, i, ctypes.byref(r))
            out.append({"name": r.name.decode(errors="surrogateescape"), "st_mode": r.st_mode,
                        "is_regular": bool(r.is_regular), "size": r.size,
                        "sha512": lib().snaphash_records_sha512_hex(h, i).decode()})
        return arch.value.decode(), out
    finally:
        lib().snaphash_records_free(h)


def numa_probe(sysfs_root, pci_bus_id):
    """The engines' topology probe on any sysfs tree -> (node, [cpus])."""
    node, n = ctypes.c_int32(), ctypes.c_size_t()
    cpus = (ctypes.c_int32 * 4096)()
    rc = lib().snaphash_numa_probe(os.fsencode(sysfs_root), pci_bus_id.encode(), ctypes.byref(node), cpus, 4096, ctypes.byref(n))
    if rc:
        raise SnaphashError(rc)
    return node.value, list(cpus[:min(n.value, 4096)])


def mode_string(st_mode):
    buf = ctypes.create_string_buffer(11)
    rc = lib().snaphash_mode_string(st_mode, buf)
    if rc:
        raise SnaphashError(rc)
    return buf.value.decode()


def mode_parse(s):
    m = ctypes.c_uint32()
    rc = lib().snaphash_mode_parse(s.encode(), ctypes.byref(m))
    if rc:
        raise SnaphashError(rc)
    return m.value


def lpt_assign(lens, nshards):
    import numpy as np
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    out = np.empty(len(lens), dtype=np.int32)
    rc = lib().snaphash_lpt_assign(lens.ctypes.data, len(lens), nshards, out.ctypes.data)
    if rc:
        raise SnaphashError(rc)
    return out
