#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04p; mkdir -p $O
python -m pytest tests/test_gpu_early_insert.py tests/test_driver_fuzz.py tests/test_gpu_train_loop.py -m gpu -x -q 2>&1 | grep -v "resource_tracker\|cache\[rtype\]\|KeyError: ./psm\|Traceback (most" | tail -25 > $O/tests.txt
timeout 120 python tools/soak_early_insert.py --seconds 15 --unmasked 2>&1 | tail -1 >> $O/tests.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 5"
for i in 1 2 3; do
  $B --selector prioritized 2>/dev/null | grep '^{' > $O/prio_predict_$i.json
  EMB_PREDICT_ROWS=0 $B --selector prioritized 2>/dev/null | grep '^{' > $O/prio_nopredict_$i.json
done
for i in 1 2; do
  $B 2>/dev/null | grep '^{' > $O/ppo_predict_$i.json
  EMB_PREDICT_ROWS=0 $B 2>/dev/null | grep '^{' > $O/ppo_nopredict_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('sustained') or {}
  print(f.split('/')[-1].ljust(26), 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'us/step', s.get('ms_per_step'), 'fence', s.get('closing_fence_us'))
PY
cat $O/tests.txt
