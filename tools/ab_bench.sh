#!/bin/bash
# A/B of bench.py variants on ONE box (boxes differ by +-10 %): alternating runs.
#   tools/ab_bench.sh OUTDIR "ENV_A" "ENV_B" [rounds] [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$(cd "$R" && mkdir -p "$1" && cd "$1" && pwd); A=$2; B=$3; N=${4:-2}; shift 4
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    env $E python "$R/bench.py" --no-cpu-baseline --no-context --no-dreamer-leg --sustained-seconds 4 "$@" 2>/dev/null | grep '^{' > "$O/ab_${v}_$i.json"
    python - "$O/ab_${v}_$i.json" "$v$i [$E]" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d['sustained']
print(sys.argv[2], 'value', d['value'], 'sustained', s['env_steps_per_s'], 'us/step', round(s['ms_per_step'] * 1e3, 2),
      'gather', s['gather_avg_us'])
PY
  done
done
