"""The oracle (oracle/sha512_oracle.c) pinned against the reference's own known
answers and against hashlib (OpenSSL) on FIPS 180-4 boundary lengths.  CPU only."""
import hashlib
import json
import os
import stat

import pytest

from conftest import GOLDEN
import trees


def test_reference_kats(oracle):
    kats = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))["kats"]
    assert len(kats) == 4
    for k in kats:
        assert oracle.sha512(k["input_utf8"].encode()).hex() == k["sha512"], k["source"]


def test_sha512sum_file_kat(oracle, tmp_path):
    # helpers/helpers_test.go:167-176 TestSha512sum
    p = tmp_path / "test.txt"
    p.write_bytes(b"x")
    assert oracle.sha512sum(str(p)) == (
        "a4abd4448c49562d828115d13a1fccea927f52b4d5459297f8b43e42da89238b"
        "c13626e43dcb38ddb082488927ec904fb42057443983e88585179d50551afe62")


def test_sha512sum_missing_file(oracle, tmp_path):
    with pytest.raises(OSError):
        oracle.sha512sum(str(tmp_path / "nope"))


def test_boundary_digests(oracle):
    vec = json.load(open(os.path.join(GOLDEN, "boundary_digests.json")))["vectors"]
    assert len(vec) >= 30
    for r in vec:
        data = oracle.fill_synthetic(r["length"], r["file_index"]).tobytes()
        assert data[:8].hex() == r["first8"]
        assert oracle.sha512(data).hex() == r["sha512"], r["length"]


def test_against_hashlib_random(oracle):
    rnd = os.urandom(70000)
    for n in list(range(0, 300)) + [1 << 12, 32768, 32769, 65535, 70000]:
        assert oracle.sha512(rnd[:n]) == hashlib.sha512(rnd[:n]).digest(), n


def test_streaming_file_matches_oneshot(oracle, tmp_path):
    data = os.urandom(3 * 32768 + 77)  # spans several 32 KiB io.Copy reads
    p = tmp_path / "f"
    p.write_bytes(data)
    assert oracle.sha512sum(str(p)) == hashlib.sha512(data).hexdigest()


def test_golden_hashes_yaml(oracle, tmp_path):
    # snappy/hashes_test.go:57-104 TestBuildCreateDebianHashesSimple, byte for byte
    build, tar = trees.make_simple_tree(str(tmp_path))
    want = open(os.path.join(GOLDEN, "hashes_simple.yaml"), "rb").read()
    assert oracle.hashes_yaml(build, tar) == want
    oracle.write_hashes(build, tar)
    out = os.path.join(build, "DEBIAN", "hashes.yaml")
    assert open(out, "rb").read() == want
    assert stat.S_IMODE(os.stat(out).st_mode) == 0o644


def test_mode_strings(oracle):
    # snappy/hashes_test.go:30-33: dir with 0644 -> drw-r--r--
    assert oracle.mode_string(stat.S_IFDIR | 0o644) == "drw-r--r--"
    assert oracle.mode_string(stat.S_IFREG | 0o644) == "frw-r--r--"
    assert oracle.mode_string(stat.S_IFLNK | 0o777) == "lrwxrwxrwx"
    assert oracle.mode_string(stat.S_IFREG | 0o4755) == "frwxr-xr-x"  # setuid dropped
    for bad in (stat.S_IFIFO, stat.S_IFSOCK, stat.S_IFCHR, stat.S_IFBLK):
        with pytest.raises(ValueError):
            oracle.mode_string(bad | 0o644)


def test_walk_order_and_debian_prefix(oracle, tmp_path):
    b = tmp_path / "b"
    for d in ("a", "DEBIAN", "DEBIAN-extra", "sub/DEBIAN"):
        (b / d).mkdir(parents=True)
    for f in ("a/x", "a-b", "DEBIANfoo", "DEBIAN/control", "DEBIAN-extra/y", "sub/DEBIAN/kept", "debian"):
        (b / f).write_bytes(f.encode())
    tar = tmp_path / "t"
    tar.write_bytes(b"")
    y = oracle.hashes_yaml(str(b), str(tar)).decode()
    names = [l[len("- name: "):] for l in y.splitlines() if l.startswith("- name: ")]
    # per-directory byte-wise pre-order: 'a' < 'a-b' as names, but a's children come right after a
    assert names == ["a", "a/x", "a-b", "debian", "sub", "sub/DEBIAN", "sub/DEBIAN/kept"]


def test_fifo_aborts_the_pass(oracle, tmp_path):
    b = tmp_path / "b"
    b.mkdir()
    (b / "ok").write_bytes(b"1")
    os.mkfifo(str(b / "pipe"))
    tar = tmp_path / "t"
    tar.write_bytes(b"")
    with pytest.raises(oracle.OracleError) as e:
        oracle.hashes_yaml(str(b), str(tar))
    assert e.value.code == oracle.EMODE


def test_unsafe_name_refused(oracle, tmp_path):
    b = tmp_path / "b"
    b.mkdir()
    (b / "0o17").write_bytes(b"1")  # a string or an integer depending on the Go release: not restated, refused
    tar = tmp_path / "t"
    tar.write_bytes(b"")
    with pytest.raises(oracle.OracleError) as e:
        oracle.hashes_yaml(str(b), str(tar))
    assert e.value.code == oracle.EUNSAFE


def test_empty_tree(oracle, tmp_path):
    b = tmp_path / "b"
    b.mkdir()
    tar = tmp_path / "t"
    tar.write_bytes(b"")
    y = oracle.hashes_yaml(str(b), str(tar)).decode()
    assert y.endswith("files: []\n") and y.startswith("archive-sha512: cf83e135")


def test_missing_archive_fails_first(oracle, tmp_path):
    b = tmp_path / "b"
    b.mkdir()
    with pytest.raises(oracle.OracleError) as e:
        oracle.hashes_yaml(str(b), str(tmp_path / "missing.tar.gz"))
    assert e.value.code == oracle.EIO
