"""ORACLE -- test infrastructure only.

ctypes binding of oracle/sha512_oracle.c (the CPU restatement of
snappy/build.go:216-270 + helpers/helpers.go:187-201).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

OK, EIO, EMODE, EUNSAFE, ENOMEM = 0, -1, -2, -3, -4


def build(force=False):
    src = os.path.join(_HERE, "sha512_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_sha512.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.oracle_sha512.restype = None
        L.oracle_sha512_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_size_t, ctypes.c_void_p]
        L.oracle_sha512_batch.restype = None
        L.oracle_sha512sum.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.oracle_sha512sum.restype = ctypes.c_int
        L.oracle_mode_string.argtypes = [ctypes.c_uint, ctypes.c_char_p]
        L.oracle_mode_string.restype = ctypes.c_int
        L.oracle_hashes_yaml.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p),
                                         ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
        L.oracle_hashes_yaml.restype = ctypes.c_int
        L.oracle_write_hashes.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.oracle_write_hashes.restype = ctypes.c_int
        L.oracle_free.argtypes = [ctypes.c_void_p]
        L.oracle_free.restype = None
        L.oracle_files_equal.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.oracle_files_equal.restype = ctypes.c_int
        L.oracle_dir_updated.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p),
                                         ctypes.POINTER(ctypes.c_size_t)]
        L.oracle_dir_updated.restype = ctypes.c_int
        L.oracle_fill_synthetic.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.oracle_fill_synthetic.restype = None
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code, errno_=0):
        self.code, self.errno = code, errno_
        super().__init__("oracle error %d (errno %d)" % (code, errno_))


def sha512(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(64)
    lib().oracle_sha512(data, len(data), out)
    return out.raw


def sha512_batch(base, offsets, lens):
    """base: numpy uint8 array; offsets/lens: numpy uint64 arrays -> (n,64) uint8."""
    import numpy as np
    n = len(offsets)
    out = np.empty((n, 64), dtype=np.uint8)
    lib().oracle_sha512_batch(base.ctypes.data, offsets.ctypes.data, lens.ctypes.data, n, out.ctypes.data)
    return out


def sha512sum(path: str) -> str:
    """helpers.Sha512sum: hexdigest of a file; raises OSError like the Go error return."""
    hexbuf = ctypes.create_string_buffer(129)
    e = lib().oracle_sha512sum(os.fsencode(path), hexbuf)
    if e:
        raise OSError(e, os.strerror(e), path)
    return hexbuf.value.decode()


def mode_string(st_mode: int) -> str:
    buf = ctypes.create_string_buffer(11)
    if lib().oracle_mode_string(st_mode, buf) != 0:
        raise ValueError("Unknown file mode %o" % st_mode)
    return buf.value.decode()


def hashes_yaml(build_dir: str, data_tar: str) -> bytes:
    """writeHashes minus the file write."""
    p = ctypes.c_void_p()
    n = ctypes.c_size_t()
    en = ctypes.c_int()
    rc = lib().oracle_hashes_yaml(os.fsencode(build_dir), os.fsencode(data_tar), ctypes.byref(p),
                                  ctypes.byref(n), ctypes.byref(en))
    if rc:
        raise OracleError(rc, en.value)
    try:
        return ctypes.string_at(p.value, n.value)
    finally:
        lib().oracle_free(p)


def write_hashes(build_dir: str, data_tar: str) -> None:
    en = ctypes.c_int()
    rc = lib().oracle_write_hashes(os.fsencode(build_dir), os.fsencode(data_tar), ctypes.byref(en))
    if rc:
        raise OracleError(rc, en.value)


def fill_synthetic(length: int, file_index: int):
    import numpy as np
    a = np.empty(length, dtype=np.uint8)
    if length:
        lib().oracle_fill_synthetic(a.ctypes.data, length, file_index)
    return a


def files_equal(a: str, b: str) -> bool:
    """helpers.FilesAreEqual (helpers/cmp.go:31-60)."""
    return bool(lib().oracle_files_equal(os.fsencode(a), os.fsencode(b)))


def dir_updated(dir_a: str, dir_b: str, pfx: str = ""):
    """helpers.DirUpdated (helpers/cmp.go:88-114) -> set of names (the Go map's keys)."""
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    lib().oracle_dir_updated(os.fsencode(dir_a), os.fsencode(dir_b), pfx.encode(), ctypes.byref(p), ctypes.byref(n))
    try:
        text = ctypes.string_at(p.value).decode()
    finally:
        lib().oracle_free(p)
    out = set(text.splitlines())
    assert len(out) == n.value
    return out
