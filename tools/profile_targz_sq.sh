#!/bin/bash
# SQ counters of the fused Build pass's kernels (what bounds deflate_chunks_kernel: issue, LDS or memory waits?).
# usage: tools/profile_targz_sq.sh <outdir-under-gpurun_out> [kind] [MiB]
set -o pipefail
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 "$GRAFT_REPO_ROOT/tools/targz_bench.py" "${2:-text}" "${3:-512}" > "$OUT/pmc_sq.log" 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmc_sq2" -- python3 "$GRAFT_REPO_ROOT/tools/targz_bench.py" "${2:-text}" "${3:-512}" > "$OUT/pmc_sq2.log" 2>&1 || exit 2
find "$OUT" -name "*.csv" -size +8M -delete
exit 0
