mkdir -p gpurun_out/c9
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stu -o st -- python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --unmasked-env-actions > $R/gpurun_out/c9/bench_unmasked_under_rocprof.json 2>/dev/null
cp /tmp/stu/st_kernel_stats.csv $R/gpurun_out/c9/kernel_stats_bench_unmasked.csv
for i in 1 2 3; do
  python $R/bench.py --host-envs --parallel-envs --no-cpu-baseline --no-context --no-dreamer-leg --sustained-seconds 5 --steps 10000 2>/dev/null | grep '^{' > $R/gpurun_out/c9/hostenvs_$i.json
done
python $R/tools/profile_host_step.py 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/c9/profile_host_step.txt
python - <<PY
import json
for i in (1,2,3):
  d=json.loads(open('$R/gpurun_out/c9/hostenvs_%d.json'%i).read()); s=d['sustained']
  print('hostenvs', i, d['value'], s['env_steps_per_s'], round(s['ms_per_step']*1e3,1))
PY
head -8 $R/gpurun_out/c9/kernel_stats_bench_unmasked.csv | cut -c1-200
cat $R/gpurun_out/c9/profile_host_step.txt
