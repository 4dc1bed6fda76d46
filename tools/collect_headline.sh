#!/bin/bash
# The headline subset of tools/collect_profiles.sh (run from the repo root on the GPU box):
# the default line, the driver's short form, rocprofv3 --stats of the same default command and
# of the configs[2] workload.    tools/collect_headline.sh r04h
TAG=${1:-r05h}
R=$(pwd); O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" 2>/dev/null | grep '^{' > "$O/bench_default.json"
for i in 1 2 3; do python "$R/bench.py" --steps 20 --warmup 5 2>/dev/null | grep '^{' > "$O/bench_steps20_$i.json"; done
python "$R/bench.py" --workload dreamer --context-only --steps 5000 --sustained-seconds 3 --no-dreamer-leg 2>/dev/null | grep '^{' > "$O/bench_dreamer.json"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o st -- \
  python "$R/bench.py" --no-cpu-baseline --no-dreamer-leg > "$O/bench_under_rocprof.json" 2>"$O/stats_bench.log"
cp /tmp/st/st_kernel_stats.csv "$O/kernel_stats_bench.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st2 -o st -- \
  python "$R/bench.py" --workload dreamer --context-only --steps 5000 --sustained-seconds 0 --no-cpu-baseline > /dev/null 2>&1
cp /tmp/st2/st_kernel_stats.csv "$O/kernel_stats_dreamer.csv"
echo "wrote $O"
