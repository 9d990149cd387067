"""Tree builders shared by the tests."""
import os

from snappy_amd import synthetic


def make_simple_tree(root):
    """The 5-entry tree of the reference's TestBuildCreateDebianHashesSimple
    (snappy/hashes_test.go:57-87): returns (build_dir, data_tar)."""
    build = os.path.join(root, "build")
    os.makedirs(os.path.join(build, "DEBIAN"), mode=0o755)
    os.chmod(os.path.join(build, "DEBIAN"), 0o755)
    _write(os.path.join(build, "DEBIAN", "bar"), b"", 0o644)   # debian dir is ignored
    _write(os.path.join(build, "foo"), b"", 0o644)             # regular files are looked at
    os.makedirs(os.path.join(build, "bin"), mode=0o755)         # normal subdirs are supported
    os.chmod(os.path.join(build, "bin"), 0o755)
    _write(os.path.join(build, "bin", "bar"), b"bar\n", 0o644)
    os.symlink("/dsafdsafsadf", os.path.join(build, "broken-link"))  # symlinks are supported
    tar_dir = os.path.join(root, "tar")
    os.makedirs(tar_dir)
    data_tar = os.path.join(tar_dir, "data.tar.gz")
    _write(data_tar, b"", 0o644)
    return build, data_tar


def _write(path, data, mode):
    with open(path, "wb") as f:
        f.write(data)
    os.chmod(path, mode)


def make_synthetic_tree(root, sizes, first_index=0):
    """d{i//100:04d}/f{i:06d}.bin tree with SplitMix64 content;
    the last size is the data.tar stand-in.  Returns (build_dir, data_tar)."""
    build = os.path.join(root, "build")
    os.makedirs(build, mode=0o755)
    os.chmod(build, 0o755)
    n = len(sizes) - 1
    for i in range(n):
        p = os.path.join(build, synthetic.file_name(first_index + i))
        d = os.path.dirname(p)
        if not os.path.isdir(d):
            os.makedirs(d, mode=0o755)
            os.chmod(d, 0o755)
        _write(p, synthetic.file_bytes(int(sizes[i]), first_index + i), 0o644)
    data_tar = os.path.join(root, "data.tar.gz")
    _write(data_tar, synthetic.file_bytes(int(sizes[n]), first_index + n), 0o644)
    return build, data_tar
