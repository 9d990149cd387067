#!/bin/bash
# One-shot latency of the CLI on a small package (300 files, ~20 MiB): what a single `snappy build` pays -- process start, HIP and ctx
# initialisation, staging buffers -- beside the pass itself.  usage (repo root, GPU box): bash tools/cli_latency_probe.sh
set -e
T=$(mktemp -d /dev/shm/snaphash_cli_XXXX)
python3 - "$T" <<'PY'
import os, sys, numpy as np
root = sys.argv[1]
rng = np.random.default_rng(21)
sizes = np.minimum(16 << 20, np.maximum(1, rng.lognormal(np.log(20 << 10), 2.0, size=300))).astype(np.int64)
pool = open(sys.executable, "rb").read() * 8
os.makedirs(os.path.join(root, "build", "DEBIAN"))
for i, sz in enumerate(sizes):
    d = os.path.join(root, "build", "d%03d" % (i // 100)); os.makedirs(d, exist_ok=True)
    off = int(rng.integers(0, len(pool) - int(sz) - 1))
    open(os.path.join(d, "f%05d" % i), "wb").write(pool[off:off + int(sz)])
print("tree of %.1f MiB" % (sizes.sum() / 2**20))
PY
for i in 1 2 3; do
  a=$(date +%s%N); SNAPHASH_TRACE_TARGZ=1 snappy_amd/bin/snaphash build "$T/build" "$T/data.tar.gz" > /dev/null; b=$(date +%s%N)
  echo "snaphash build: $(( (b - a) / 1000000 )) ms wall"
done
for i in 1 2; do
  a=$(date +%s%N); snappy_amd/bin/snaphash tree "$T/build" "$T/data.tar.gz" > /dev/null; b=$(date +%s%N)
  echo "snaphash tree: $(( (b - a) / 1000000 )) ms wall"
done
rm -rf "$T"
