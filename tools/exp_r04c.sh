R=$(pwd); O=$R/gpurun_out/r04d; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "callbacks_may_keep" 2>&1 | grep -v "resource_tracker\|cache\[rtype\]\|KeyError\|Traceback" | tail -30 > $O/tests_callbacks.txt
cd /tmp
python $R/tools/profile_step.py > $O/profile_step_compiled.txt 2>&1
EMB_PURE_PYTHON=1 python $R/tools/profile_step.py > $O/profile_step_plain.txt 2>&1
EMB_HOST_PROFILE=1 python $R/bench.py --no-cpu-baseline --no-dreamer-leg --no-context --sustained-seconds 3 --streams 1 2> $O/host_profile_native.txt | cut -c1-300
cat $O/tests_callbacks.txt; head -8 $O/profile_step_compiled.txt; head -8 $O/profile_step_plain.txt; grep "emb host profile" $O/host_profile_native.txt
