#!/bin/bash
# rocprofv3 evidence for bench.py's workload: kernel trace + stats, then HBM counters in
# their own passes (gpurun refuses --pmc combined with trace domains).  Run on the GPU box:
#   bash tools/profile_bench.sh <out_dir_under_gpurun_out> [bench args...]
set -o pipefail
OUT=${1:-gpurun_out/prof}; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || exit 2
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$tag" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> "$OUT/pmc_$tag.err" || echo "pmc $c failed"
done
find "$OUT" -type f | head -40
cat "$OUT/bench_trace.json"
