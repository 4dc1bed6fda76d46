#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 5"
for i in 1 2; do
  for se in 4 16 64; do
    $B --stamp-every $se 2>/dev/null | grep '^{' > $O/ppo_se${se}_$i.json
    $B --workload dreamer --steps 5000 --stamp-every $se 2>/dev/null | grep '^{' > $O/dreamer_se${se}_$i.json
  done
  EMB_BENCH_NO_TIMER=1 $B 2>/dev/null | grep '^{' > $O/ppo_notimer_$i.json
  EMB_BENCH_NO_TIMER=1 $B --workload dreamer --steps 5000 2>/dev/null | grep '^{' > $O/dreamer_notimer_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('sustained') or {}
  print(f.split('/')[-1].ljust(26), 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'gather', s.get('gather_avg_us'), s.get('gather_launches'), 'wb', s.get('writeback_avg_us'))
PY
