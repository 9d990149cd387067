#!/bin/bash
# rocprofv3 kernel trace of the fused Build pass (row f3): per-kernel time of deflate_chunks_kernel,
# deflate_compact_kernel and the SHA-512 kernels.  usage: tools/profile_targz.sh <outdir-under-gpurun_out> [kind] [MiB]
set -o pipefail
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/tools/targz_bench.py" "${2:-text}" "${3:-1024}" > "$OUT/trace.log" 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$GRAFT_REPO_ROOT/tools/targz_bench.py" "${2:-text}" "${3:-1024}" > "$OUT/pmc_fetch.log" 2>&1 || exit 2
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$GRAFT_REPO_ROOT/tools/targz_bench.py" "${2:-text}" "${3:-1024}" > "$OUT/pmc_write.log" 2>&1 || exit 3
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 "$GRAFT_REPO_ROOT/tools/targz_bench.py" "${2:-text}" "${3:-1024}" > "$OUT/pmc_sq.log" 2>&1 || exit 4
# round 5: how busy the vector units are and how full their waves (lane utilisation = THREAD_CYCLES / (ACTIVE_INST x 64)); a counter
# this build of rocprofv3 does not know fails this pass only
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_valu" -- python3 "$GRAFT_REPO_ROOT/tools/targz_bench.py" "${2:-text}" "${3:-1024}" > "$OUT/pmc_valu.log" 2>&1 || echo "pmc_valu pass failed" >> "$OUT/pmc_valu.log"
find "$OUT" -name "*.csv" -size +8M -delete
exit 0
