mkdir -p gpurun_out/c4
R=$GRAFT_REPO_ROOT
(HIP_FORCE_DEV_KERNARG=0 tools/build/kernarg_lab; HIP_FORCE_DEV_KERNARG=1 tools/build/kernarg_lab) > gpurun_out/c4/kernarg_lab.txt 2>&1
cd /tmp
D="--workload dreamer --context-only --steps 4000 --sustained-seconds 2 --no-dreamer-leg --no-cpu-baseline --no-context"
for v in "" "EMB_LAB_SCATTER_PER_CU=8" "EMB_LAB_SCATTER_PER_CU=6" "EMB_LAB_SCATTER_PER_CU=2" "EMB_LAB_SCATTER_FLAT=1"; do
  env $v python $R/bench.py $D 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); wb=d['writeback']; s=d['sustained']
print('$v', 'sust', s['env_steps_per_s'], 'gather', s['gather_avg_us'], 'wb', wb['avg_launch_us'], wb['frac'], wb['kernel'])" >> $R/gpurun_out/c4/scatter_lab.txt
done
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "wide_observations or parallel_env or host_mode or callbacks_may" 2>&1 | grep -v -E "^(RCCL|HIP.ver|ROCm|Hostname|Librccl)" | tail -15 > gpurun_out/c4/t_host.txt
cd /tmp
for i in 1 2; do python $R/bench.py --host-envs --parallel-envs --no-cpu-baseline --no-context --no-dreamer-leg --sustained-seconds 5 2>gpurun_out_err.txt | grep '^{' > $R/gpurun_out/c4/hostenvs_$i.json; done
python $R/bench.py --host-envs --parallel-envs --envs-per-worker 1 --no-cpu-baseline --no-context --no-dreamer-leg --sustained-seconds 5 2>/dev/null | grep '^{' > $R/gpurun_out/c4/hostenvs_k1.json
cd $R
( time EMB_RCCL_LIB=$(python -c "import importlib.util,sys; s=importlib.util.spec_from_file_location('b','tests/fake_rccl/build.py'); m=importlib.util.module_from_spec(s); s.loader.exec_module(m); print(m.build())") timeout 900 python bench.py --gpus 8 --backend gloo --no-cpu-baseline --steps 4000 --warmup 200 --sustained-seconds 3 > gpurun_out/c4/rehearsal_short.json 2> gpurun_out/c4/rehearsal_short.err ) 2> gpurun_out/c4/rehearsal_short.time
cat gpurun_out/c4/kernarg_lab.txt gpurun_out/c4/scatter_lab.txt gpurun_out/c4/t_host.txt gpurun_out/c4/rehearsal_short.time
