"""helpers.Sha512sum, GPU-backed (reference helpers/helpers.go:187-201).

Same names, argument meaning and error behaviour as the Go functions: a path
in, a lowercase hex digest out, OSError (Go: the error return of os.Open /
io.Copy) on any I/O failure.  Sha512sumBatch is the batched seam the cgo shim
adds: one call per writeHashes pass instead of one per file.
"""
from ._lib import Context

_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def Sha512sumBatch(paths, ctx=None):
    """[path] -> [hexdigest]; the first failing file raises and no digest is returned."""
    ctx = ctx or default_context()
    return [d.hex() for d in ctx.sha512_files(list(paths))]


def Sha512sum(infile, ctx=None):
    """Sha512sum returns the sha512 of the given file as a hexdigest."""
    return Sha512sumBatch([infile], ctx)[0]


def FilesAreEqual(a, b, ctx=None):
    """FilesAreEqual compares the two files' contents and returns whether they are the same
    (reference helpers/cmp.go:31-60); the bytes are compared on the GPU."""
    return (ctx or default_context()).files_equal([(a, b)])[0]


def DirUpdated(dirA, dirB, pfx, ctx=None):
    """DirUpdated compares two directories, and returns which files present in both have been
    updated, with the given prefix prepended.  Subdirectories are ignored (helpers/cmp.go:88-114)."""
    return (ctx or default_context()).dir_updated(dirA, dirB, pfx)
