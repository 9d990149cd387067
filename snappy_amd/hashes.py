"""The hashes.yaml pass (reference snappy/build.go:216-270, snappy/hashes.go).

writeHashes(buildDir, dataTar) -- exactly the reference function: archive digest,
    walk, per-file SHA-512 on the GPU, DEBIAN/hashes.yaml written 0644.
getHashes(buildDir, dataTar)   -- the same minus the file write (returns the
    YAML bytes); the name BASELINE.json's north_star uses.
Verify(instDir, yaml, dataTar=None) -- the inverse (absent upstream; hook point
    snappy/click.go:970): returns None or (kind, name) of the first mismatch.
"""
import os
import stat

from . import _lib
from .helpers import default_context


class yamlFileMode:
    """snappy/hashes.go:25-88 -- os.FileMode <-> "frw-r--r--"."""

    def __init__(self, st_mode):
        self.mode = st_mode

    def MarshalYAML(self):
        return _lib.mode_string(self.mode)

    @classmethod
    def UnmarshalYAML(cls, s):
        return cls(_lib.mode_parse(s))

    def __eq__(self, o):
        return isinstance(o, yamlFileMode) and (stat.S_IFMT(self.mode), stat.S_IMODE(self.mode) & 0o777) == \
            (stat.S_IFMT(o.mode), stat.S_IMODE(o.mode) & 0o777)


def getHashes(buildDir, dataTar, ctx=None):
    return (ctx or default_context()).tree(buildDir, dataTar)


def writeHashes(buildDir, dataTar, ctx=None):
    (ctx or default_context()).write_hashes(buildDir, dataTar)


def Verify(instDir, yaml_bytes=None, dataTar=None, ctx=None):
    if yaml_bytes is None:  # installed layout: <inst>/meta/hashes.yaml (snappy/click.go:330-338)
        with open(os.path.join(instDir, "meta", "hashes.yaml"), "rb") as f:
            yaml_bytes = f.read()
    return (ctx or default_context()).verify(instDir, yaml_bytes, dataTar)
