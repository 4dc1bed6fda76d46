#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04m; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_early_insert.py tests/test_driver_fuzz.py -m gpu -x -q -k "parallel or host_envs or fuzz or callbacks" 2>&1 | grep -v "resource_tracker\|cache\[rtype\]\|KeyError: ./psm\|Traceback (most" | tail -6 > $O/tests_hostenv.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 5"
for i in 1 2; do
  $B 2>/dev/null | grep '^{' > $O/ppo_$i.json
  EMB_X_SKIP_PUBLISH_LAUNCH=1 $B 2>/dev/null | grep '^{' > $O/ppo_skip_publish_$i.json
  $B --host-envs --parallel-envs 2>/dev/null | grep '^{' > $O/hostenvs_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('sustained') or {}
  print(f.split('/')[-1].ljust(26), 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'us/step', s.get('ms_per_step'), 'fence', s.get('closing_fence_us'))
PY
cat $O/tests_hostenv.txt
