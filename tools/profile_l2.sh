#!/bin/bash
# L2 / HBM-side counters of one workload (separate --pmc passes).
#   bash tools/profile_l2.sh <out_dir_under_gpurun_out> <python script> [args...]
set -o pipefail
OUT=${1:-gpurun_out/l2}; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $c --output-format csv -d "$OUT/p$i" -- python3 "$@" > "$OUT/p$i.out" 2> "$OUT/p$i.err" || echo "pass $i ($c) failed"
done
python3 tools/summarize_pmc.py "$OUT"
