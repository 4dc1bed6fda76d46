#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 4 --steps 20000"
for v in "512,2 1" "256,4 0" "256,4 1" "512,2 0" "256,3 0" "256,2 0" "1024,1 0"; do
  set -- $v
  EMB_SPAN_VARIANT=4,3,$1 EMB_SPAN_BALANCE=$2 $B --streams 1 2>/dev/null | grep '^{' > $O/ppo_v_${1/,/_}_b$2.json
  EMB_SPAN_VARIANT=4,1,$1,160,3 EMB_SPAN_BALANCE=$2 $B --workload dreamer --steps 4000 --streams 2 2>/dev/null | grep '^{' > $O/dreamer_v_${1/,/_}_b$2.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
  except Exception as e:
    print(f, 'ERR', e); continue
  s=d.get('sustained') or {}
  print(f.split('/')[-1].ljust(28), 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'us/step', s.get('ms_per_step'), 'gather', s.get('gather_avg_us'), (d.get('roofline') or {}).get('avg_launch_us'), 'wb', s.get('writeback_avg_us'))
PY
