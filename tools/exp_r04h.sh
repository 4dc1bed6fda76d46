#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04j; mkdir -p $O; rm -f $O/shapes2.txt
cd /tmp && export TMPDIR=/tmp
for v in "512,2 1" "256,4 0" "256,5 0" "256,4 1"; do
  set -- $v
  echo "== EMB_SPAN_VARIANT=4,3,$1 EMB_SPAN_BALANCE=$2" >> $O/shapes2.txt
  EMB_SPAN_VARIANT=4,3,$1 EMB_SPAN_BALANCE=$2 python $R/tools/bench_gather.py --batches 1,2,4,8,12,16,24,32,128 --tight --iters 100 2>&1 | grep "tight" >> $O/shapes2.txt
done
cat $O/shapes2.txt
