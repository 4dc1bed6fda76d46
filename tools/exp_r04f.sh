#!/bin/bash
# Span mover shapes at B=16 (S0 rows), tight loop, three repetitions each.
R=$(pwd); O=$R/gpurun_out/r04h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for v in "512,2 1" "512,2 0" "256,4 0" "256,4 1" "256,3 0" "256,6 0" "256,8 0" "512,3 0" "512,4 0" "1024,2 0"; do
  set -- $v
  echo "== rep $rep EMB_SPAN_VARIANT=4,3,$1 EMB_SPAN_BALANCE=$2" >> $O/gather_shapes.txt
  EMB_SPAN_VARIANT=4,3,$1 EMB_SPAN_BALANCE=$2 python $R/tools/bench_gather.py --batches 16,64 --tight --iters 150 2>&1 | grep "tight" >> $O/gather_shapes.txt
done
done
python - <<PY
import re,collections
rows=collections.defaultdict(list); cur=None
for l in open('$O/gather_shapes.txt'):
  if l.startswith('=='): cur=' '.join(l.split()[3:])
  else:
    m=re.search(r'B=\s*(\d+) kernel\s+([\d.]+)', l)
    if m: rows[(cur,int(m.group(1)))].append(float(m.group(2)))
for k in sorted(rows): print(k, rows[k], 'min', min(rows[k]))
PY
