"""snappy_amd -- MI355X-native SHA-512 integrity pass behind snappy's hashes.yaml seam.

Host-side mirror (Python, for tests/bench) of the two Go entry points the
reference has on this path, over the C ABI of libsnaphash.so:

  helpers.Sha512sum / Sha512sumBatch    reference helpers/helpers.go:187-201
  hashes.writeHashes / getHashes / Verify   reference snappy/build.go:216-270

Hashing runs in hand-written HIP kernels on gfx950 only; nothing here
computes a digest on the CPU.
"""
from . import _lib
from ._lib import Context, SnaphashError
from .helpers import Sha512sum, Sha512sumBatch, FilesAreEqual, DirUpdated
from .hashes import writeHashes, getHashes, Verify, yamlFileMode

__all__ = ["Context", "SnaphashError", "Sha512sum", "Sha512sumBatch", "FilesAreEqual", "DirUpdated", "writeHashes", "getHashes", "Verify",
           "yamlFileMode", "_lib"]
