#!/bin/bash
# A/B of the learner's stream on ONE box: one stream, a plain second stream, CU-masked second
# streams of several widths; alternating rounds.   tools/exp_learner_cus.sh OUTDIR [rounds] [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/$1; N=${2:-2}; shift 2
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  for v in s1 s2_0 s2_32 s2_64 s2_96 s2_128; do
    case $v in
      s1) F="--streams 1";;
      *) F="--streams 2 --learner-cus ${v#s2_}";;
    esac
    python "$R/bench.py" --no-cpu-baseline --no-context --no-dreamer-leg --sustained-seconds 4 $F "$@" 2>/dev/null | grep '^{' > "$O/${v}_$i.json"
    python - "$O/${v}_$i.json" "$v round $i $*" <<'PY'
import json, sys
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  s = d['sustained']
  print(sys.argv[2], 'value', d['value'], 'sustained', s['env_steps_per_s'], 'us/step', round(s['ms_per_step'] * 1e3, 2),
        'gather', s['gather_avg_us'], 'fence', s['closing_fence_us'], flush=True)
except Exception as e:
  print(sys.argv[2], 'FAILED', e, flush=True)
PY
  done
done
