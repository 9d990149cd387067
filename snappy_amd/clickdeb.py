"""clickdeb.tarCreate, GPU-backed (reference clickdeb/deb.go:261-344).

Same name, argument meaning and error behaviour as the Go function: tarCreate(tarname, sourceDir, fn) walks
sourceDir, asks fn(path) for every regular file, symlink and directory (False leaves it out; None keeps all),
writes members "./<relative path>" owned by root through gzip into tarname, and raises on the first error.
Only ".gz" here: the reference's ".xz" branch shells out to an external tool.  Test/bench harness, like
helpers.py and hashes.py: the product is the C ABI.
"""
from .helpers import default_context


def tarCreate(tarname, sourceDir, fn=None, ctx=None):
    """-> the 64-byte SHA-512 of the archive written (the Go function returns only the error)."""
    _, digest = (ctx or default_context()).tar_create_fn(tarname, sourceDir, fn)
    return digest
