#!/bin/bash
# Reproduce profiles/rNN_* on the GPU box (run from the repo root):
#   tools/collect_profiles.sh r01
# Counters are collected in their own passes (never together with a trace domain
# other than --kernel-trace); FETCH_SIZE and WRITE_SIZE do not fit one pass.
set -e
TAG=${1:-r01}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" > "$O/bench_default.json" 2>/dev/null
python "$R/bench.py" --workload dreamer --steps 5000 2>/dev/null | grep '^{' > "$O/bench_dreamer.json"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o st -- \
  python "$R/bench.py" --no-cpu-baseline > "$O/stats_bench.log" 2>&1
cp /tmp/st/st_kernel_stats.csv "$O/kernel_stats.csv"
for counter in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $counter --kernel-trace --output-format csv -d /tmp/p_$counter -o p -- \
    python "$R/bench.py" --steps 300 --no-cpu-baseline > /dev/null 2>&1
  python "$R/tools/summarize_pmc.py" /tmp/p_$counter/p_counter_collection.csv > "$O/pmc_$counter.csv"
done
rocprofv3 --kernel-trace --output-format csv -d /tmp/km -o km -- \
  python "$R/tools/bench_kernels.py" > /dev/null 2>&1
cp /tmp/km/km_kernel_trace.csv "$O/kernels_micro_trace.csv" 2>/dev/null || true
echo "wrote $O"
