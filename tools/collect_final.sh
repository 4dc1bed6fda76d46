# Everything profiles/rNN_* is made from, on one GPU box (run from the repo root):
#   tools/collect_final.sh r06
TAG=${1:-r06}
set -x
mkdir -p gpurun_out/$TAG
timeout 1500 tools/collect_profiles.sh $TAG > gpurun_out/$TAG/collect.log 2>&1
R=$(pwd); O=$R/gpurun_out/$TAG
cd /tmp
for i in 1 2 3; do python $R/bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_steps20_$i.json; done
python $R/bench.py --selector prioritized --no-cpu-baseline --no-dreamer-leg 2>/dev/null | grep '^{' > $O/bench_selectorprioritized.json
# alternating A/Bs on this box: the env's action form (value = masked, the reference's) and the shipped selector
python $R/tools/ab_runs.py $O/ab_masked.txt 3 "masked (default, driver.py:72-75)::" "unmasked + reset::--unmasked-env-actions"
python $R/tools/ab_runs.py $O/ab_prioritized.txt 3 "uniform (default)::" "prioritized (ppo/configs.yaml:42)::--selector prioritized"
python $R/bench.py --host-envs --parallel-envs --no-cpu-baseline --no-dreamer-leg 2>/dev/null | grep '^{' > $O/bench_hostenvsparallelenvs.json
python $R/tools/bench_index.py > $O/bench_index.txt 2>&1
# the reference's own perf loops (per-step add, sample(1), Driver over Dummy envs) and the insert-route fuzz
python $R/tools/perf_reference_scripts.py --seconds 3 > $O/perf_reference_scripts.txt 2>/dev/null
timeout 300 python $R/tools/fuzz_add_paths.py --seeds 1000 --steps 600 2>&1 | tail -1 > $O/fuzz_add_paths.txt
HIP_FORCE_DEV_KERNARG=0 python $R/tools/profile_host_step.py > $O/profile_host_step.txt 2>&1
# timelines (queues, overlap, the sequence of a window) of both workloads
for w in ppo dreamer; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$w -o tr -- python $R/bench.py --workload $w $([ $w = dreamer ] && echo --context-only) --steps 2000 --warmup 100 --sustained-seconds 0 --no-cpu-baseline --no-context --no-dreamer-leg --capacity 100000 > /dev/null 2>&1
  python $R/tools/trace_overlap.py $(find /tmp/tl_$w -name "*kernel_trace.csv" | head -1) 0.7 > $O/timeline_$w.txt 2>&1
done
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O/gpu_tests.txt
