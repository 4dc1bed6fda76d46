mkdir -p gpurun_out/c3
F="grep -v -E ^(RCCL|HIP.ver|ROCm|Hostname|Librccl)"
python -m pytest tests/test_gpu_direct_comm_ranks.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q 2>&1 | $F | tail -15 > gpurun_out/c3/t1.txt
tools/exp_learner_cus.sh gpurun_out/c3/masked 1 > gpurun_out/c3/ab_masked.txt 2>&1
cd /tmp
for i in 1 2; do
python $GRAFT_REPO_ROOT/bench.py --workload dreamer --context-only --steps 5000 --sustained-seconds 3 --no-dreamer-leg --no-cpu-baseline --no-context 2>/dev/null | grep '^{' > $GRAFT_REPO_ROOT/gpurun_out/c3/dreamer_$i.json
python $GRAFT_REPO_ROOT/bench.py --selector prioritized --no-dreamer-leg --no-cpu-baseline --no-context --sustained-seconds 4 2>/dev/null | grep '^{' > $GRAFT_REPO_ROOT/gpurun_out/c3/prio_$i.json
python $GRAFT_REPO_ROOT/bench.py --no-dreamer-leg --no-cpu-baseline --no-context --sustained-seconds 4 2>/dev/null | grep '^{' > $GRAFT_REPO_ROOT/gpurun_out/c3/uni_$i.json
done
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bench_launcher.py -x -q --durations=8 2>&1 | $F | tail -40 > gpurun_out/c3/t2.txt
cat gpurun_out/c3/t1.txt gpurun_out/c3/ab_masked.txt; tail -15 gpurun_out/c3/t2.txt
