mkdir -p gpurun_out/c6
R=$GRAFT_REPO_ROOT
F="grep -v -E ^(RCCL|HIP.ver|ROCm|Hostname|Librccl)"
python -m pytest tests/test_gpu_parity.py -x -q -k "wide_observations or parallel_env or host_mode or callbacks_may" 2>&1 | $F | tail -8 > gpurun_out/c6/t_host.txt
python -m pytest tests/test_gpu_early_insert.py tests/test_gpu_train_loop.py -x -q 2>&1 | $F | tail -5 >> gpurun_out/c6/t_host.txt
cd /tmp
for i in 1 2; do
 for v in "EMB_BENCH_UPLOAD_GROUPS=1 EMB_BENCH_WORKER_SPIN_US=0" "EMB_BENCH_UPLOAD_GROUPS=1 EMB_BENCH_WORKER_SPIN_US=400" "EMB_BENCH_UPLOAD_GROUPS=4 EMB_BENCH_WORKER_SPIN_US=0" "EMB_BENCH_UPLOAD_GROUPS=4 EMB_BENCH_WORKER_SPIN_US=400" "EMB_BENCH_UPLOAD_GROUPS=8 EMB_BENCH_WORKER_SPIN_US=400" "EMB_BENCH_UPLOAD_GROUPS=2 EMB_BENCH_WORKER_SPIN_US=400"; do
  env $v python $R/bench.py --host-envs --parallel-envs --no-cpu-baseline --no-context --no-dreamer-leg --sustained-seconds 4 --steps 10000 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['sustained']
print('$v round $i', 'value', d['value'], 'sustained', s['env_steps_per_s'], 'us/step', round(s['ms_per_step']*1e3,1), 'K', d['config'].get('envs_per_worker'))" >> $R/gpurun_out/c6/ab_hostenvs.txt
 done
done
env EMB_BENCH_UPLOAD_GROUPS=4 EMB_BENCH_WORKER_SPIN_US=400 python $R/bench.py --host-envs --parallel-envs --envs-per-worker 1 --no-cpu-baseline --no-context --no-dreamer-leg --sustained-seconds 4 --steps 10000 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['sustained']
print('K=1 groups 4 spin 400', 'value', d['value'], 'sustained', s['env_steps_per_s'], 'us/step', round(s['ms_per_step']*1e3,1), 'K', d['config'].get('envs_per_worker'))" >> $R/gpurun_out/c6/ab_hostenvs.txt
(echo "== defaults (4 pieces by kernel, store, spin 400)"; python $R/tools/profile_host_step.py; echo "== round 5 form (one DMA copy, copies down, semaphores)"; EMB_BENCH_UPLOAD_GROUPS=1 EMB_BENCH_ACTS_BY_STORE=0 EMB_BENCH_WORKER_SPIN_US=0 python $R/tools/profile_host_step.py) 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/c6/profile_host_step.txt
cd $R
cat gpurun_out/c6/t_host.txt gpurun_out/c6/ab_hostenvs.txt gpurun_out/c6/profile_host_step.txt
