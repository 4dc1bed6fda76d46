set -x
mkdir -p gpurun_out/r03
timeout 1500 tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1
R=$(pwd); O=$R/gpurun_out/r03
cd /tmp
for i in 1 2 3; do python $R/bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_steps20_$i.json; done
python $R/bench.py --selector prioritized --no-cpu-baseline --no-dreamer-leg 2>/dev/null | grep '^{' > $O/bench_selectorprioritized.json
python $R/bench.py --host-envs --parallel-envs --no-cpu-baseline --no-dreamer-leg 2>/dev/null | grep '^{' > $O/bench_hostenvsparallelenvs.json
python $R/tools/bench_index.py > $O/bench_index.txt 2>&1
HIP_FORCE_DEV_KERNARG=0 python $R/tools/profile_host_step.py > $O/profile_host_step.txt 2>&1
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O/gpu_tests.txt
