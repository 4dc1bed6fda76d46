#!/bin/bash
# Prioritized selector: zero-on-sample draw with contiguous child masses in the sample tree,
# one range per draw, zero-window shortcut -- library before / after, alternating runs.
R=$(pwd); O=$R/gpurun_out/r04v; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "prior or select or golden or mixture" 2>&1 | grep -E "passed|failed|error" > $O/tests.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --selector prioritized --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 5"
for i in 1 2 3; do
  EMB_LIB_PATH=$R/tools/build/libembodied_hip_before.so $B 2>/dev/null | grep '^{' > $O/before_$i.json
  $B 2>/dev/null | grep '^{' > $O/after_$i.json
done
EMB_HOST_PROFILE=1 $B > /dev/null 2> $O/host.txt
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
  d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get('sustained') or {}
  print(f.split('/')[-1].ljust(18), 'value', d['value'], 'sustained', s.get('env_steps_per_s'), 'ms/step', s.get('ms_per_step'))
PY
cat $O/tests.txt; grep "sample: index\|add: index\|peek check" $O/host.txt
