#!/bin/bash
# SQ counters of one workload in separate rocprofv3 --pmc passes (no trace domains mixed in).
#   bash tools/profile_pmc.sh <out_dir_under_gpurun_out> <python script> [args...]
set -o pipefail
OUT=${1:-gpurun_out/pmc}; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $c --output-format csv -d "$OUT/p$i" -- python3 "$@" > "$OUT/p$i.out" 2> "$OUT/p$i.err" || echo "pass $i ($c) failed"
done
python3 tools/summarize_pmc.py "$OUT"
