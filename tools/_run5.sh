mkdir -p gpurun_out/c5
R=$GRAFT_REPO_ROOT
F="grep -v -E ^(RCCL|HIP.ver|ROCm|Hostname|Librccl)"
python -m pytest tests/test_gpu_bench_launcher.py -x -q --durations=6 -k "rehearsal or dreamer_workload or checks_the_native or c10d_keeps" 2>&1 | $F | tail -25 > gpurun_out/c5/t_launcher.txt
cd /tmp
for i in 1 2; do
 for v in "EMB_BENCH_UPLOAD_GROUPS=1 EMB_BENCH_ACTS_BY_STORE=0" "EMB_BENCH_UPLOAD_GROUPS=4 EMB_BENCH_ACTS_BY_STORE=0" "EMB_BENCH_UPLOAD_GROUPS=1 EMB_BENCH_ACTS_BY_STORE=1" "EMB_BENCH_UPLOAD_GROUPS=4 EMB_BENCH_ACTS_BY_STORE=1" "EMB_BENCH_UPLOAD_GROUPS=8 EMB_BENCH_ACTS_BY_STORE=1"; do
  env $v python $R/bench.py --host-envs --parallel-envs --no-cpu-baseline --no-context --no-dreamer-leg --sustained-seconds 4 --steps 10000 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['sustained']
print('$v round $i', 'value', d['value'], 'sustained', s['env_steps_per_s'], 'us/step', round(s['ms_per_step']*1e3,1), 'K', d['config'].get('envs_per_worker'))" >> $R/gpurun_out/c5/ab_hostenvs.txt
 done
done
(echo "== pieces 4, store"; python $R/tools/profile_host_step.py; echo "== one copy each (round 5)"; EMB_BENCH_UPLOAD_GROUPS=1 EMB_BENCH_ACTS_BY_STORE=0 python $R/tools/profile_host_step.py) 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/c5/profile_host_step.txt
cd $R
cat gpurun_out/c5/t_launcher.txt gpurun_out/c5/ab_hostenvs.txt gpurun_out/c5/profile_host_step.txt
