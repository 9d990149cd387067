#!/bin/bash
# SQ counters of deflate_chunks_kernel on the sources corpus, for the shipped build and the 8-byte-extension build (make narrow).
# usage: tools/profile_deflate_sources.sh <outdir-under-gpurun_out>
set -o pipefail
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$GRAFT_REPO_ROOT/tools/deflate_sources_probe.py"
for v in wide narrow; do
  if [ $v = narrow ]; then export SNAPHASH_LIB="$GRAFT_REPO_ROOT/snappy_amd/variants/libsnaphash_narrow.so"; else unset SNAPHASH_LIB; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${v}_trace" -- python3 "$P" > "$OUT/${v}_trace.log" 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${v}_pmc" -- python3 "$P" > "$OUT/${v}_pmc.log" 2>&1 || exit 2
done
find "$OUT" -name "*.csv" -size +8M -delete
exit 0
