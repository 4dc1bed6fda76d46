#!/bin/bash
# PPO with the learner on its own stream: does a narrower gather (fewer
# workgroups per CU) or a low-priority learner stream let the env step's
# kernels run beside it?
R=$(pwd); O=$R/gpurun_out/r04q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 5"
for i in 1 2; do
  $B 2>/dev/null | grep '^{' > $O/s1_$i.json
  $B --streams 2 2>/dev/null | grep '^{' > $O/s2_w4_$i.json
  EMB_SPAN_VARIANT=4,3,256,2 $B --streams 2 2>/dev/null | grep '^{' > $O/s2_w2_$i.json
  EMB_SPAN_VARIANT=4,3,256,1 $B --streams 2 2>/dev/null | grep '^{' > $O/s2_w1_$i.json
  EMB_BENCH_LEARNER_PRIORITY=1 $B --streams 2 2>/dev/null | grep '^{' > $O/s2_w4_low_$i.json
  EMB_BENCH_LEARNER_PRIORITY=1 EMB_SPAN_VARIANT=4,3,256,2 $B --streams 2 2>/dev/null | grep '^{' > $O/s2_w2_low_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('sustained') or {}
  r=d.get('roofline') or {}
  print(f.split('/')[-1].ljust(26), 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'us/step', s.get('ms_per_step'), 'gather_us', r.get('launch_us'), r.get('frac'))
PY
