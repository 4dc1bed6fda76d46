#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 6 --steps 20000"
for i in 1 2 3; do
  $B 2>/dev/null | grep '^{' > $O/ppo_default_$i.json
  EMB_SPAN_VARIANT=4,3,256,4 EMB_SPAN_BALANCE=0 $B 2>/dev/null | grep '^{' > $O/ppo_256x4_$i.json
  $B --workload dreamer --steps 4000 2>/dev/null | grep '^{' > $O/dreamer_default_$i.json
  EMB_SPAN_VARIANT=4,1,256,4,160,3 EMB_SPAN_BALANCE=0 $B --workload dreamer --steps 4000 2>/dev/null | grep '^{' > $O/dreamer_256x4_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('sustained') or {}
  print(f.split('/')[-1].ljust(26), 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'gather', (d.get('roofline') or {}).get('headline_region',{}).get('avg_launch_us'), s.get('gather_avg_us'), 'wb', s.get('writeback_avg_us'))
PY
