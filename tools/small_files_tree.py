#!/usr/bin/env python3
"""A tree of many small files and no big one (default 100 000 x 8 KiB) -> hashes.yaml: how many fill threads serve it
best?  Every open + close takes the process's one descriptor-table lock (tools/openat_probe.cpp), so more threads can
be slower.  Each configuration in a process of its own (SNAPHASH_COPY_THREADS is read once).
usage: tools/small_files_tree.py [n=100000] [KiB=8]"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(build):
    tar = os.path.join(os.path.dirname(build), "data.tar.gz")
    from snappy_amd import Context, _lib
    rows = {}
    for name, kw in (("GPU only", dict(flags=_lib.FLAG_GPU_ONLY)), ("default", dict(flags=0))):
        with Context(**kw) as c:
            best = None
            for _ in range(4):
                t0 = time.perf_counter(); y = c.tree(build, tar); dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, c.stats(), c.stats_ex())
            rows[name] = {"ms": best[0] * 1e3, "kernel_ms": best[1]["kernel_ms"], "h2d_ms": best[1]["h2d_ms"], "host_streams": best[2]["host_streams"]}
    import hashlib
    rows["yaml"] = hashlib.sha256(y).hexdigest()[:12]
    print(json.dumps(rows))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    kib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tmp = tempfile.mkdtemp(prefix="snaphash_small_", dir="/dev/shm")
    try:
        build = os.path.join(tmp, "build")
        blob = os.urandom(kib << 10)
        for i in range(n):
            d = os.path.join(build, "d%04d" % (i // 100))
            if i % 100 == 0:
                os.makedirs(d)
            with open(os.path.join(d, "f%06d.bin" % i), "wb") as f:
                f.write(blob[i % 97:] + blob[:i % 97])
        with open(os.path.join(tmp, "data.tar.gz"), "wb") as f:
            f.write(b"archive stand-in")
        print("# %d files of %d KiB (%.2f GiB) in %d directories, best of 4 passes" % (n, kib, n * kib / 2**20, (n + 99) // 100), flush=True)
        for threads in (None, 2, 4, 6, 8, 12):
            env = dict(os.environ)
            env.pop("SNAPHASH_COPY_THREADS", None)
            if threads:
                env["SNAPHASH_COPY_THREADS"] = str(threads)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", build], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            if r.returncode:
                print("fill threads %s: FAILED %s" % (threads, r.stderr.decode()[-300:]))
                continue
            j = json.loads(r.stdout.decode().strip().splitlines()[-1])
            print("fill threads %-8s GPU only %.1f ms (kernels %.0f, h2d %.0f)   default %.1f ms (%d streams on host threads)   yaml %s" % (
                threads or "(engine)", j["GPU only"]["ms"], j["GPU only"]["kernel_ms"], j["GPU only"]["h2d_ms"], j["default"]["ms"], j["default"]["host_streams"], j["yaml"]), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
