#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04e; mkdir -p $O
python -m pytest tests/test_gpu_early_insert.py tests/test_driver_fuzz.py tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_train_loop.py -m gpu -x -q 2>&1 | grep -v "resource_tracker\|cache\[rtype\]\|KeyError: '/psm\|Traceback (most" | tail -15 > $O/tests.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 5"
for i in 1 2; do
  $B --streams 2 2>$O/ppo_s2_$i.err | grep '^{' > $O/ppo_s2_$i.json
  $B --streams 1 2>/dev/null | grep '^{' > $O/ppo_s1_$i.json
done
$B --workload dreamer --steps 5000 --streams 2 2>/dev/null | grep '^{' > $O/dreamer_s2.json
for s in 1 2; do
  EMB_BENCH_TRACE_STEPS=1 python $R/bench.py --steps 40 --warmup 50 --sustained-seconds 0 --no-cpu-baseline --no-context --no-dreamer-leg --streams $s 2>&1 >/dev/null | grep "per-step" > $O/trace_steps_s$s.txt
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
  except Exception as e:
    print(f, 'ERR', e); continue
  s=d.get('sustained') or {}
  print(f.split('/')[-1], 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'us/step', s.get('ms_per_step'), 'gather', s.get('gather_avg_us'), 'wb', s.get('writeback_avg_us'), 'fence', s.get('closing_fence_us'))
PY
cat $O/tests.txt $O/trace_steps_s*.txt
