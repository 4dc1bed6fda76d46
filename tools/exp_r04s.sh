#!/bin/bash
# obs-stack + early-insert launch: (frame blocks + 1, n) grid against the flat, XCD-even one.
R=$(pwd); O=$R/gpurun_out/r04s; mkdir -p $O
python -m pytest tests/test_gpu_early_insert.py tests/test_driver_fuzz.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|error' > $O/tests.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 5"
for i in 1 2; do
  EMB_LIB_PATH=$R/tools/build/libembodied_hip_before.so $B 2>/dev/null | grep '^{' > $O/before_$i.json
  $B 2>/dev/null | grep '^{' > $O/after_$i.json
done
for v in before after; do
  L=""; [ $v = before ] && L=$R/tools/build/libembodied_hip_before.so
  EMB_LIB_PATH=$L rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 0 --steps 20000 > /dev/null 2>&1
  f=$(find $O/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v" >> $O/kernels.txt
  python - "$f" >> $O/kernels.txt <<PY
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:6]: print(r[0][:60].ljust(62), r[1], r[3])
PY
  rm -rf $O/prof_$v
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('sustained') or {}
  print(f.split('/')[-1].ljust(26), 'value', d['value'], 'sust', s.get('env_steps_per_s'), 'us/step', s.get('ms_per_step'))
PY
cat $O/tests.txt; grep -v '^"' $O/kernels.txt
