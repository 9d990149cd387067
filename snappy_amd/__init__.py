"""snappy_amd -- MI355X-native SHA-512 integrity pass behind snappy's hashes.yaml seam.

Host-side mirror (Python, for tests/bench) of the two Go entry points the
reference has on this path, over the C ABI of libsnaphash.so:

  helpers.Sha512sum / Sha512sumBatch    reference helpers/helpers.go:187-201
  hashes.writeHashes / getHashes / Verify   reference snappy/build.go:216-270

The HIP kernels (gfx950 only) hash what many streams in parallel make fast.  Since
ABI 4 every call is planned, and the library's OWN host SHA-512 (csrc/hostsha.cpp,
hostsha_x8.cpp -- part of libsnaphash.so, never the test suite's checker) takes the
streams a lone 44 MB/s GPU stream would make slower than the reference's single
goroutine; Context.stats_ex() says which bytes went where, flags=FLAG_GPU_ONLY keeps
every byte on the GPU.  There is no fallback: without a gfx950 device Context() raises.
Nothing in this Python package computes a digest itself.
"""
from . import _lib
from ._lib import Context, SnaphashError
from .helpers import Sha512sum, Sha512sumBatch, FilesAreEqual, DirUpdated
from .hashes import writeHashes, getHashes, Verify, yamlFileMode

__all__ = ["Context", "SnaphashError", "Sha512sum", "Sha512sumBatch", "FilesAreEqual", "DirUpdated", "writeHashes", "getHashes", "Verify",
           "yamlFileMode", "_lib"]
