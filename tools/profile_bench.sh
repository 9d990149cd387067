#!/bin/bash
# Profiles bench.py's N=1 run with rocprofv3 on the GPU box: kernel trace + stats, then
# separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950).
# usage: tools/profile_bench.sh <outdir-under-gpurun_out> [bench args...]
set -o pipefail
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$B" --steps 5 --warmup 2 --cpu-seconds 0 --legs off "$@" > "$OUT/bench_under_trace.log" 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$B" --steps 2 --warmup 1 --cpu-seconds 0 --legs off "$@" > "$OUT/pmc_fetch.log" 2>&1 || exit 2
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$B" --steps 2 --warmup 1 --cpu-seconds 0 --legs off "$@" > "$OUT/pmc_write.log" 2>&1 || exit 3
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 "$B" --steps 2 --warmup 1 --cpu-seconds 0 --legs off "$@" > "$OUT/pmc_sq.log" 2>&1 || exit 4
# keep only the small summaries (the raw traces can be large)
find "$OUT" -name "*.csv" -size +8M -delete
exit 0
