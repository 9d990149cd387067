import os, sys, time, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
t0=time.perf_counter()
import torch
from snappy_amd import Context, _lib
import trees
t1=time.perf_counter()
with tempfile.TemporaryDirectory() as tmp:
    build, tar = trees.make_synthetic_tree(tmp, [1000, 50000, 200000, 3, 1<<20, 4096]*4)
    _lib.lib()
    t2=time.perf_counter()
    c=Context(flags=0)
    t3=time.perf_counter()
    y=c.tree(build, tar)
    t4=time.perf_counter()
    y=c.tree(build, tar)
    t5=time.perf_counter()
    out=os.path.join(tmp,"o.tar.gz")
    c.tar_create(out, build, build+"/DEBIAN", with_hashes=True)
    t6=time.perf_counter()
    c.tar_create(out, build, build+"/DEBIAN", with_hashes=True)
    t7=time.perf_counter()
    c.close()
    print("import torch+lib %.0f ms; load lib %.0f ms; snaphash_init %.1f ms; first tree (24 files, 5 MB) %.1f ms; second tree %.1f ms; first tar_create %.1f ms; second %.1f ms" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,(t5-t4)*1e3,(t6-t5)*1e3,(t7-t6)*1e3))
