#!/bin/bash
# Round-4 experiment batch A (run from the repo root on the GPU box).
R=$(pwd); O=$R/gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_scan_golden.py tests/test_gpu_output_pool.py -m gpu -x -q 2>&1 | tail -5 > $O/tests_new.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "callbacks_may_keep or ring_of_one or device_driver or golden" 2>&1 | tail -5 >> $O/tests_new.txt
python -m pytest tests/test_gpu_host_kernargs.py -m gpu -x -q -k "pace_rule" 2>&1 | tail -5 >> $O/tests_new.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context > $O/bench_ppo.json 2>$O/bench_ppo.err
for v in "4,3,512,2" "2,3,512,2" "4,3,256,4" "2,3,256,4" "4,3,1024,1"; do
  for bal in 1 0; do
    echo "== EMB_SPAN_VARIANT=$v EMB_SPAN_BALANCE=$bal" >> $O/gather_variants.txt
    EMB_SPAN_VARIANT=$v EMB_SPAN_BALANCE=$bal python $R/tools/bench_gather.py --batches 16 --tight --iters 100 2>&1 | grep "tight\|variant" >> $O/gather_variants.txt
  done
done
for i in 1 2; do
  python $R/bench.py --workload dreamer --steps 5000 --sustained-seconds 3 --no-dreamer-leg --no-cpu-baseline --no-context 2>/dev/null | grep '^{' > $O/dreamer_new_$i.json
  EMB_SPAN_VARIANT=4,3 EMB_BENCH_LAMBDA_PAIR=0 python $R/bench.py --workload dreamer --steps 5000 --sustained-seconds 3 --no-dreamer-leg --no-cpu-baseline --no-context 2>/dev/null | grep '^{' > $O/dreamer_old_$i.json
  EMB_SPAN_VARIANT=4,3 python $R/bench.py --workload dreamer --steps 5000 --sustained-seconds 3 --no-dreamer-leg --no-cpu-baseline --no-context 2>/dev/null | grep '^{' > $O/dreamer_pair_only_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
  except Exception as e:
    print(f, 'ERR', e); continue
  s=d.get('sustained') or {}
  print(f.split('/')[-1], 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'us/step', s.get('ms_per_step'), 'gather', s.get('gather_avg_us'), 'wb', s.get('writeback_avg_us'), 'fence', s.get('closing_fence_us'), (d.get('roofline') or {}).get('kernel'))
PY
cat $O/tests_new.txt; cat $O/gather_variants.txt
