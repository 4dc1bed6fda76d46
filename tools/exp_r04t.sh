#!/bin/bash
# HSA_ENABLE_INTERRUPT=0 (the runtime polls completion signals instead of sleeping on an
# interrupt): what does it do to the waits of the path -- the closing fence of the driver's
# short form, the per-step wait for the actions with host envs?
R=$(pwd); O=$R/gpurun_out/r04t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context"
for i in 1 2 3; do
  $B --steps 20 --warmup 5 --sustained-seconds 2 2>/dev/null | grep '^{' > $O/short_irq_$i.json
  HSA_ENABLE_INTERRUPT=0 $B --steps 20 --warmup 5 --sustained-seconds 2 2>/dev/null | grep '^{' > $O/short_poll_$i.json
done
for i in 1 2; do
  $B --host-envs --parallel-envs --sustained-seconds 4 2>/dev/null | grep '^{' > $O/host_irq_$i.json
  HSA_ENABLE_INTERRUPT=0 $B --host-envs --parallel-envs --sustained-seconds 4 2>/dev/null | grep '^{' > $O/host_poll_$i.json
done
$B --sustained-seconds 4 2>/dev/null | grep '^{' > $O/ppo_irq_1.json
HSA_ENABLE_INTERRUPT=0 $B --sustained-seconds 4 2>/dev/null | grep '^{' > $O/ppo_poll_1.json
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('sustained') or {}
  print(f.split('/')[-1].ljust(22), 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'ms/step', d.get('ms_per_step'), 'fence', d.get('closing_fence_us'), s.get('closing_fence_us'))
PY
