#!/bin/bash
# Refresh of the headline lines after the last code changes + the whole GPU suite.
R=$(pwd); O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_default.json 2>/dev/null
for i in "" _1 _2 _3; do python $R/bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_steps20$i.json; done
python $R/bench.py --selector prioritized --no-cpu-baseline --no-dreamer-leg 2>/dev/null | grep '^{' > $O/bench_selectorprioritized.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o st -- python $R/bench.py --no-cpu-baseline --no-dreamer-leg > $O/bench_under_rocprof.json 2>/dev/null
cp /tmp/st/st_kernel_stats.csv $O/kernel_stats_bench.csv
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O/gpu_tests.txt
cat $O/gpu_tests.txt
python - <<PY
import json
for f in ['bench_default','bench_steps20','bench_steps20_1','bench_steps20_2','bench_steps20_3','bench_selectorprioritized']:
  d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); s=d['sustained']; r=d['roofline']
  print(f, d['value'], s['env_steps_per_s'], s['ms_per_step'], r['avg_launch_us'], r['frac'], r['launches'])
PY
head -6 $O/kernel_stats_bench.csv | cut -c1-120
