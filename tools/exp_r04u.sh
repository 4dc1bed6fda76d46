#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
EMB_HOST_PROFILE=1 python $R/bench.py --selector prioritized --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 4 > $O/prio.json 2> $O/prio_host.txt
grep "emb host profile" $O/prio_host.txt
python - <<PY
import json
d=json.loads(open('$O/prio.json').read().strip().splitlines()[-1]); s=d.get('sustained') or {}
print('value', d['value'], 'sust', s.get('env_steps_per_s'), d.get('publishes'), s.get('closing_fence_us'))
PY
