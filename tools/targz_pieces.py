#!/usr/bin/env python3
"""How a slot of the fused Build pass is cut into pieces for the compressor (targz.inc gz_process_slot): piece size, a short
first piece (the archive digest starts sooner), halving last pieces (less left for the digest after the last kernel).
One tree (text, 1 GiB, 1 MiB files), every configuration in a process of its own (the knobs are read once), several
passes each.  usage: tools/targz_pieces.py [MiB=1024] [passes=6] [kind=text]"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = [
    {},
    {"SNAPHASH_ZPIECE": "512"},
    {"SNAPHASH_ZFIRST": "128"},
    {"SNAPHASH_ZFIRST": "256"},
    {"SNAPHASH_ZFIRST": "248"},
    {"SNAPHASH_ZFIRST": "256", "SNAPHASH_ZTAPER": "128"},
    {"SNAPHASH_ZFIRST": "512"},
    {"SNAPHASH_ZPIECE": "512", "SNAPHASH_ZFIRST": "128"},
    {"SNAPHASH_ZPIECE": "768", "SNAPHASH_ZFIRST": "192"},
    {},
]


def child(build, out, passes):
    from snappy_amd import Context
    with Context() as c:
        rows = []
        for rep in range(passes):
            t0 = time.perf_counter()
            y, dig = c.tar_create(out, build, build + "/DEBIAN", with_hashes=True)
            dt = time.perf_counter() - t0
            zs, st = c.targz_stats(), c.stats()
            rows.append({"s": dt, "deflate_ms": zs["deflate_ms"], "sha_ms": st["kernel_ms"], "gz": zs["gz_bytes"], "tar": zs["tar_bytes"]})
        import hashlib
        ok = hashlib.sha512(open(out, "rb").read()).digest() == dig
        print(json.dumps({"rows": rows, "digest_ok": ok, "yaml_sha": hashlib.sha256(y).hexdigest()[:16],
                          "gz_sha": hashlib.sha256(open(out, "rb").read()).hexdigest()[:16]}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    total_mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    kind = sys.argv[3] if len(sys.argv) > 3 else "text"
    import numpy as np
    from snappy_amd import synthetic
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="snaphash_pieces_", dir=base)
    try:
        build = os.path.join(tmp, "build")
        os.makedirs(os.path.join(build, "DEBIAN"))
        open(os.path.join(build, "DEBIAN", "control"), "w").write("Package: bench\n")
        rng = np.random.default_rng(5)
        words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
        idx = rng.zipf(1.3, size=(4 << 20) // 5 + 16) % 2000
        tblock = b" ".join(words[int(i)] for i in idx)[:4 << 20]
        fsize = 1 << 20
        for i in range(total_mib):
            p = os.path.join(build, synthetic.file_name(i))
            os.makedirs(os.path.dirname(p), exist_ok=True)
            if kind == "text" or (kind == "mixed" and i % 2 == 0):
                off = int(rng.integers(0, len(tblock) - 1))
                data = (tblock[off:] + tblock)[:fsize]
            else:
                data = synthetic.file_bytes(fsize, i)
            with open(p, "wb") as f:
                f.write(data)
        out = os.path.join(tmp, "data.tar.gz")
        print("# %d MiB of %s in 1 MiB files, %d passes per configuration (first pass dropped): wall per pass min / median, "
              "compressor kernels per pass" % (total_mib, kind, passes), flush=True)
        ref = None
        for cfg in CONFIGS:
            env = dict(os.environ)
            for k in ("SNAPHASH_ZPIECE", "SNAPHASH_ZFIRST", "SNAPHASH_ZTAPER"):
                env.pop(k, None)
            env.update(cfg)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", build, out, str(passes)], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            name = " ".join("%s=%s" % (k[9:], v) for k, v in sorted(cfg.items())) or "default (ZPIECE=1024)"
            if r.returncode != 0:
                print("%-44s FAILED rc=%d %s" % (name, r.returncode, r.stderr.decode()[-300:]), flush=True)
                continue
            j = json.loads(r.stdout.decode().strip().splitlines()[-1])
            rows = j["rows"][1:]
            t = sorted(x["s"] for x in rows)
            d = sorted(x["deflate_ms"] for x in rows)
            same = ""
            if ref is None:
                ref = (j["gz_sha"], j["yaml_sha"])
            same = "same bytes" if (j["gz_sha"], j["yaml_sha"]) == ref else "DIFFERENT BYTES"
            trace = [l for l in r.stderr.decode().splitlines() if "snaphash targz" in l]
            print("%-44s %.1f / %.1f ms   deflate %.1f ms   digest ok %s, %s%s" % (
                name, t[0] * 1e3, t[len(t) // 2] * 1e3, d[len(d) // 2], j["digest_ok"], same,
                ("   [" + " | ".join(t.split("targz: ")[1] for t in trace[-2:]) + "]") if trace else ""), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
