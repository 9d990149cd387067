"""File names that are not plain YAML scalars (snappy/build.go:249-264 marshals ANY name through yaml.v2).

PARITY UNPINNED: no fixture of the reference holds a quoted or folded name (snappy/hashes_test.go:89-103 is plain
names only), and gopkg.in/yaml.v2 @ 49c95bdc is not in the reference tree.  What is checked here is therefore not
parity but three weaker things, stated as such:
  1. round trip -- the text the product writes is read back to the same name by the repository's own parser AND by
     PyYAML's reader (BaseLoader: syntax only, no type resolution);
  2. second opinion on the scalar analysis -- for names that do not resolve to another YAML type, libyaml's own C
     emitter (PyYAML CSafeDumper, width 80; yaml.v2's emitterc.go is a port of that code) must write the very same
     bytes, folding included;
  3. the oracle's independent-in-language restatement (C) equals the product's (C++) byte for byte.
CPU only: the digests the emitter needs come from hashlib / the oracle, nothing here hashes through the library."""
import hashlib
import os

import pytest
import yaml

NAMES_STR = [  # stay strings under yaml.v2's resolve(): plain, single- or double-quoted by the libyaml analysis alone
    "1.txt", ".hidden", "foo bar", "icon@2x.png", "a~", "x:y", "x: y", "@foo", "-", "- a", "-a", "a #b", "a#b", "#a",
    "tab\there", "café", "quote'", 'dq"', "back\\slash", " lead", "trail ", "...", "---x", "a, b", "[x]", "{x}", "?", "? a",
    "?a", "!bang", "&a", "*a", "|a", ">a", "%a", "`a", "emoji\U0001F600", "bom﻿x", "\x01ctl", "nULL", "Yess", "+", "nbsp x",
    "long " + "word " * 30 + "end", "long  double  space " + "w " * 50 + "e", "q' " + "w " * 50 + "e", "\x02 " + "w " * 50 + "e",
    "d\\ " + "w  " * 40 + "e", "x" * 100 + " y z", "a" * 72, "a" * 73 + " b", "sp " * 26 + "x", "1.5.3", "1e", "0x", "v1.0-rc1+b2",
]
NAMES_RESOLVED = {  # yaml.v2 double-quotes what would read back as bool / null / int / float (encode.go stringv)
    "~": '"~"', "true": '"true"', "Null": '"Null"', "y": '"y"', "yes": '"yes"', "OFF": '"OFF"', "123": '"123"', "0x1F": '"0x1F"',
    "1e3": '"1e3"', "08": '"08"', "007": '"007"', "1_000": '"1_000"', "0b101": '"0b101"', "-0b11": '"-0b11"', "+0": '"+0"',
    "+.inf": '"+.inf"', ".5": '".5"', ".NaN": '".NaN"', "-1.5e-3": '"-1.5e-3"', "+Inf": '"+Inf"', "1:30": '"1:30"',
    "-12:30:00.5": '"-12:30:00.5"',
}
# resolve.go takes a scalar for a number only when strconv returns err == nil, and a value out of range IS an error
# (ErrRange): a float that overflows a double, a 0x literal or decimal of more than 64 bits that is no float either, 65 binary
# digits -- all stay strings, written plain.  Beside each the neighbour that still fits, which is quoted.
NAMES_OUT_OF_RANGE = ["1e999", "123e4567", "5e1234", "-1e400", ".5e999", "0x" + "1" * 17, "0x123456789abcdef012345", "-0x8000000000000001",
                      "+0xffffffffffffffff", "0b" + "1" * 65, "-0b" + "1" * 64, "1" * 250 + "e60", "1.7976931348623159e308",
                      # the first 100 digits of 2^1024 - 2^970 (the midpoint above the largest double), last digit raised by one
                      "1.79769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587208e308"]
NAMES_STILL_IN_RANGE = ["1e308", "1.7976931348623157e308", "1.7976931348623158e308", "1e-999", "0x" + "f" * 16, "-0x8000000000000000", "+0x7fffffffffffffff",
                        "0b" + "1" * 64, "-0b" + "1" * 63, "0b-101", "18446744073709551616", "9" * 250, "0" * 70 + "7",
                        "1.79769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207e308"]
NAMES_REFUSED = ["0o17", "<<", "0x1p-2", "+0b1", "bad\xff".encode("latin-1"), b"nl\nx"]


def _tree_with(tmp_path, name):
    d = tmp_path / "t"
    d.mkdir()
    p = os.path.join(os.fsencode(str(d)), name if isinstance(name, bytes) else os.fsencode(name))
    with open(p, "wb") as f:
        f.write(b"bar\n")
    return str(d)


def _emit(tree):
    from snappy_amd import _lib
    return _lib.emit_yaml(tree, hashlib.sha512(b"").digest(), [hashlib.sha512(b"bar\n").digest()])


def _scalar(y):
    body = y.decode().split("files:\n", 1)[1]
    return body[len("- name:"):body.index("\n  size:")]


@pytest.mark.parametrize("name", NAMES_STR)
def test_string_names_round_trip_and_match_libyaml(built_lib, tmp_path, name):
    from snappy_amd import _lib
    y = _emit(_tree_with(tmp_path, name))
    assert yaml.load(y.decode(), Loader=yaml.BaseLoader)["files"][0]["name"] == name  # PyYAML's reader
    assert _lib.parse_yaml(y)[1][0]["name"] == name                                     # the library's own reader
    if yaml.__with_libyaml__:  # libyaml's emitter: same style, same quoting, same folding
        want = yaml.dump([{"name": name}], Dumper=yaml.CSafeDumper, allow_unicode=True, width=80, default_flow_style=False)
        assert _scalar(y) == want[len("- name:"):].rstrip("\n")


@pytest.mark.parametrize("name", sorted(NAMES_RESOLVED))
def test_resolvable_names_are_double_quoted(built_lib, tmp_path, name):
    from snappy_amd import _lib
    y = _emit(_tree_with(tmp_path, name))
    assert _scalar(y) == " " + NAMES_RESOLVED[name]
    assert yaml.load(y.decode(), Loader=yaml.BaseLoader)["files"][0]["name"] == name
    assert _lib.parse_yaml(y)[1][0]["name"] == name
    assert yaml.safe_load(y.decode())["files"][0]["name"] == name  # quoted: even a resolving reader gets the string


@pytest.mark.parametrize("name", NAMES_OUT_OF_RANGE + NAMES_STILL_IN_RANGE)
def test_a_number_out_of_range_is_a_string(built_lib, oracle, tmp_path, name):
    """strconv's range errors decide the style: out of range -> !!str -> plain; the neighbour that fits -> double-quoted.
    Python's own float()/int() (correctly rounded, arbitrary precision) give the expected answer independently of
    both restatements; product (exact digit compare) and oracle (strtoull/strtod) must agree with it and each other."""
    from snappy_amd import _lib
    tree = _tree_with(tmp_path, name)
    y = _emit(tree)
    plain = name.replace("_", "")
    body = plain.lstrip("+-")
    if body.startswith(("0x", "0X")):
        v = int(plain, 16)
        fits = -2**63 <= v < 2**63 or (not plain.startswith(("+", "-")) and v < 2**64)
    elif "0b" in plain[:3]:
        neg = plain.startswith("-0b")
        v = int(plain[3 if neg else 2:], 2)
        fits = -2**63 <= v < 2**63 or (not neg and not plain[2:].startswith(("+", "-")) and 0 <= v < 2**64)
    else:
        fits = abs(float(plain)) != float("inf")
    assert _scalar(y) == (' "%s"' % name if fits else " " + name)
    assert (name in NAMES_STILL_IN_RANGE) == fits
    assert _lib.parse_yaml(y)[1][0]["name"] == name
    tar = tmp_path / "a.tar"
    tar.write_bytes(b"")
    want = oracle.hashes_yaml(tree, str(tar))
    assert y.split(b"files:\n", 1)[1] == want.split(b"files:\n", 1)[1]


@pytest.mark.parametrize("name", NAMES_REFUSED)
def test_names_outside_the_restatement_are_refused(built_lib, tmp_path, name):
    """Go-release-dependent spellings, invalid UTF-8 (yaml.v2: !!binary) and line breaks are refused, not guessed."""
    from snappy_amd import _lib
    with pytest.raises(_lib.SnaphashError) as e:
        _emit(_tree_with(tmp_path, name))
    assert e.value.code == _lib.ENAME


def test_oracle_and_product_write_the_same_yaml_for_odd_names(built_lib, oracle, tmp_path):
    from snappy_amd import _lib
    b = tmp_path / "build"
    (b / "DEBIAN").mkdir(parents=True)
    (b / "sub dir").mkdir()
    names = NAMES_STR + sorted(NAMES_RESOLVED)
    for i, n in enumerate(names):
        with open(os.path.join(str(b), n), "wb") as f:
            f.write(b"%d" % i)
        with open(os.path.join(str(b), "sub dir", n), "wb") as f:
            f.write(b"s%d" % i)
    tar = tmp_path / "data.tar.gz"
    tar.write_bytes(b"")
    want = oracle.hashes_yaml(str(b), str(tar))
    recs = _lib.walk(str(b))
    digs = [hashlib.sha512(open(r["path"], "rb").read()).digest() for r in recs if r["is_regular"]]
    got = _lib.emit_yaml(str(b), hashlib.sha512(b"").digest(), digs)
    assert got == want
    arch, parsed = _lib.parse_yaml(got)
    assert [p["name"] for p in parsed] == [r["name"] for r in recs]
    doc = yaml.load(got.decode(), Loader=yaml.BaseLoader)
    assert [f["name"] for f in doc["files"]] == [r["name"] for r in recs]
