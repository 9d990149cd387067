#!/bin/bash
# usage: tools/profile_saturation.sh <outdir-under-gpurun_out>
set -o pipefail
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$GRAFT_REPO_ROOT/tools/saturation_probe.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$P" > "$OUT/trace.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 "$P" > "$OUT/pmc_sq.log" 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_sq2" -- python3 "$P" > "$OUT/pmc_sq2.log" 2>&1 || exit 3
find "$OUT" -name "*.csv" -size +8M -delete
exit 0
