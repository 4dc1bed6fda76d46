#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04o; mkdir -p $O
python -m pytest tests/test_gpu_early_insert.py tests/test_driver_fuzz.py -m gpu -x -q 2>&1 | grep -v "resource_tracker\|cache\[rtype\]\|KeyError: ./psm\|Traceback (most" | tail -30 > $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $O/tests.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 5"
for i in 1 2 3; do
  $B 2>/dev/null | grep '^{' > $O/ppo_$i.json
done
EMB_BENCH_TRACE_STEPS=1 python $R/bench.py --steps 40 --warmup 50 --sustained-seconds 0 --no-cpu-baseline --no-context --no-dreamer-leg 2>&1 >/dev/null | grep "per-step" > $O/trace_steps.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/trp -o tr -- python $R/bench.py --steps 4000 --warmup 100 --sustained-seconds 0 --no-cpu-baseline --no-context --no-dreamer-leg --capacity 50000 > /dev/null 2>&1
python $R/tools/trace_overlap.py $(find /tmp/trp -name "*kernel_trace.csv" | head -1) 0.7 > $O/ppo_timeline.txt
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('sustained') or {}
  print(f.split('/')[-1].ljust(26), 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'us/step', s.get('ms_per_step'), 'fence', s.get('closing_fence_us'))
PY
cat $O/tests.txt $O/trace_steps.txt; head -45 $O/ppo_timeline.txt
