#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in 2 1; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$s -o tr -- python $R/bench.py --workload dreamer --steps 1500 --warmup 100 --sustained-seconds 0 --no-cpu-baseline --no-context --no-dreamer-leg --streams $s --capacity 200000 > /dev/null 2>&1
  python $R/tools/trace_overlap.py $(find /tmp/tr$s -name "*kernel_trace.csv" | head -1) 0.7 > $O/dreamer_overlap_s$s.txt
done
rocprofv3 --kernel-trace --output-format csv -d /tmp/trp -o tr -- python $R/bench.py --steps 4000 --warmup 100 --sustained-seconds 0 --no-cpu-baseline --no-context --no-dreamer-leg --capacity 50000 > /dev/null 2>&1
python $R/tools/trace_overlap.py $(find /tmp/trp -name "*kernel_trace.csv" | head -1) 0.7 > $O/ppo_timeline.txt
head -70 $O/dreamer_overlap_s2.txt; head -20 $O/dreamer_overlap_s1.txt; head -60 $O/ppo_timeline.txt
