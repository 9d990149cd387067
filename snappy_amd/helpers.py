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
