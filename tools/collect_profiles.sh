#!/bin/bash
# Reproduce profiles/rNN_* on the GPU box (run from the repo root):
#   tools/collect_profiles.sh r03
# Counters are collected in their own passes (never together with a trace domain
# other than --kernel-trace); FETCH_SIZE and WRITE_SIZE do not fit one pass.
set +e
TAG=${1:-r06}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# the headline line, un-profiled (bench.py's default: kernel arguments in host memory)
python "$R/bench.py" > "$O/bench_default.json" 2>/dev/null
# the runtime's default argument placement, for DESIGN.md 4's A/B
HIP_FORCE_DEV_KERNARG=1 python "$R/bench.py" --no-cpu-baseline --no-dreamer-leg > "$O/bench_device_kernargs.json" 2>/dev/null
python "$R/bench.py" --workload dreamer --context-only --steps 5000 --sustained-seconds 3 --no-dreamer-leg 2>/dev/null | grep '^{' > "$O/bench_dreamer.json"
python "$R/bench.py" --workload dreamer --steps 5000 --sustained-seconds 3 --no-dreamer-leg 2>/dev/null | grep '^{' > "$O/bench_dreamer_full_gather.json"
# the A/B of the early insert on this box (alternating runs; same everything else)
# the driver's short form
python "$R/bench.py" --steps 20 --warmup 5 2>/dev/null | grep '^{' > "$O/bench_steps20.json"
# --stats of the SAME default command
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o st -- \
  python "$R/bench.py" --no-cpu-baseline --no-dreamer-leg --no-context > "$O/bench_under_rocprof.json" 2>"$O/stats_bench.log"
cp /tmp/st/st_kernel_stats.csv "$O/kernel_stats_bench.csv"
HIP_FORCE_DEV_KERNARG=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st1 -o st -- \
  python "$R/bench.py" --no-cpu-baseline --no-dreamer-leg > "$O/bench_device_kernargs_under_rocprof.json" 2>/dev/null
cp /tmp/st1/st_kernel_stats.csv "$O/kernel_stats_bench_device_kernargs.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st2 -o st -- \
  python "$R/bench.py" --workload dreamer --context-only --steps 5000 --sustained-seconds 0 --no-cpu-baseline > /dev/null 2>&1
cp /tmp/st2/st_kernel_stats.csv "$O/kernel_stats_dreamer.csv"
for counter in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $counter --kernel-trace --output-format csv -d /tmp/p_$counter -o p -- \
    python "$R/bench.py" --steps 300 --sustained-seconds 0 --no-cpu-baseline --no-context --no-dreamer-leg > /dev/null 2>&1
  python "$R/tools/summarize_pmc.py" /tmp/p_$counter/p_counter_collection.csv > "$O/pmc_$counter.csv"
done
# the same two counter passes for the configs[2] gather (context-only: 60 MB per launch)
for counter in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $counter --kernel-trace --output-format csv -d /tmp/pd_$counter -o p -- \
    python "$R/bench.py" --workload dreamer --context-only --steps 300 --sustained-seconds 0 --no-cpu-baseline --no-context > /dev/null 2>&1
  python "$R/tools/summarize_pmc.py" /tmp/pd_$counter/p_counter_collection.csv > "$O/dreamer_pmc_$counter.csv"
done
HIP_FORCE_DEV_KERNARG=0 python "$R/tools/bench_gather.py" --batches 1,4,8,16,32,64,128,256 --tight > "$O/gather_sweep.txt" 2>&1
HIP_FORCE_DEV_KERNARG=1 python "$R/tools/bench_gather.py" --batches 1,4,8,16,32,64,128,256 --tight > "$O/gather_sweep_device_kernargs.txt" 2>&1
HIP_FORCE_DEV_KERNARG=0 python "$R/tools/profile_step.py" > "$O/profile_step.txt" 2>&1
HIP_FORCE_DEV_KERNARG=0 python "$R/tools/profile_train.py" 2>&1 | head -8 > "$O/profile_train.txt"
rocprofv3 --kernel-trace --output-format csv -d /tmp/km -o km -- \
  python "$R/tools/bench_kernels.py" > /dev/null 2>&1
python "$R/tools/summarize_trace.py" /tmp/km/km_kernel_trace.csv 3 > "$O/kernels_micro.csv" 2>/dev/null || true
# return scans at SURVEY 8d's large synthetic size
rocprofv3 --kernel-trace --output-format csv -d /tmp/sc -o sc -- python "$R/tools/bench_scans.py" > /dev/null 2>&1
python "$R/tools/summarize_trace.py" /tmp/sc/sc_kernel_trace.csv 20 | grep -i "kernel\|scan_rows" > "$O/scans_large.csv" || true
# write-back of image rows through the span mover
HIP_FORCE_DEV_KERNARG=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/up -o up -- \
  python "$R/tools/bench_update.py" 16 > /dev/null 2>&1
python "$R/tools/summarize_trace.py" /tmp/up/up_kernel_trace.csv 100 > "$O/update_image_rows.csv" 2>/dev/null || true
# the multi-rank code path with RCCL as the transport, one rank (what a 1-GPU box can run)
python "$R/bench.py" --force-dist --no-cpu-baseline --no-context --sustained-seconds 5 2>/dev/null \
  | grep '^{' > "$O/bench_world1_rccl.json" || true
python "$R/bench.py" --force-dist --comm c10d --no-cpu-baseline --no-context --sustained-seconds 5 2>/dev/null \
  | grep '^{' > "$O/bench_world1_rccl_c10d.json" || true
python - "$O" <<'PY'
import csv, json, sys
out = sys.argv[1]
def mean(path, needle):
  for row in csv.DictReader(l for l in open(path) if not l.startswith('#')):
    if needle in row['kernel']:
      return float(row['mean_per_dispatch']), int(row['dispatches']), row['kernel']
fetch, n, name = mean(f'{out}/pmc_FETCH_SIZE.csv', 'span_move_kernel')
write, _, _ = mean(f'{out}/pmc_WRITE_SIZE.csv', 'span_move_kernel')
json.dump({
    'kernel': name + ' (Replay.sample, B=16, L=65, S0=28255)',
    'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 300 '
               '--sustained-seconds 0 --no-cpu-baseline --no-context (two separate passes)',
    'dispatches': n,
    'fetch_size_kb_per_launch': fetch, 'write_size_kb_per_launch': write,
    'correction': 'FETCH_SIZE doubled: gfx950 counts 128-B requests at 64 B for 16 B/lane streams '
                  '(MI355X_MICROARCH.md, HBM)',
    'traffic_bytes_per_launch': int(round((2 * fetch + write) * 1024)),
    'algorithmic_bytes_per_launch': 2 * 16 * 65 * 28255,
}, open(f'{out}/pmc_gather.json', 'w'), indent=1)
# the three launches of a vectorised step: how often is a frame fetched?
step = {}
for needle, what in (('obs_stack_insert_kernel', 'obs stack + early insert: 64 frames read once, written as bf16 policy batch and as pool rows'),
                     ('publish_one_kernel', 'masked action to its pool rows and to the next step\'s action buffer'),
                     ('synth_env_kernel', 'synthetic env frames (benchmark input)'),
                     ('flat_move_kernel', 'plain insert (EMB_EARLY_INSERT=0 / first step) and narrow keys'),
                     ('obs_stack_kernel', 'plain obs stack')):
  try:
    f, n, name = mean(f'{out}/pmc_FETCH_SIZE.csv', needle)
    w, _, _ = mean(f'{out}/pmc_WRITE_SIZE.csv', needle)
    step[name] = {'what': what, 'dispatches': n, 'fetch_kb': f, 'write_kb': w,
                  'traffic_bytes_per_launch': int(round((2 * f + w) * 1024))}
  except Exception:
    pass
json.dump({'frames_bytes_per_step': 64 * 28224, 'kernels': step,
           'correction': 'FETCH_SIZE doubled (gfx950, 16 B/lane streams)'},
          open(f'{out}/pmc_step.json', 'w'), indent=1)
try:
  fetch, n, name = mean(f'{out}/dreamer_pmc_FETCH_SIZE.csv', 'span_move_kernel_indirect<true')
  write, _, _ = mean(f'{out}/dreamer_pmc_WRITE_SIZE.csv', 'span_move_kernel_indirect<true')
  line = json.loads(open(f'{out}/bench_dreamer.json').read().strip().splitlines()[-1])
  algorithmic = line['roofline']['bytes_per_launch']     # 2 * B * L * (bytes per step of all keys)
  json.dump({
      'kernel': name + ' (Replay.sample, configs[2]: B=16, L=65 frames + K=1 step of the latents)',
      'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --workload dreamer '
                 '--context-only --steps 300 --sustained-seconds 0 --no-cpu-baseline --no-context (two separate passes)',
      'dispatches': n,
      'fetch_size_kb_per_launch': fetch, 'write_size_kb_per_launch': write,
      'correction': 'FETCH_SIZE doubled: gfx950 counts 128-B requests at 64 B for 16 B/lane streams '
                    '(MI355X_MICROARCH.md, HBM)',
      'traffic_bytes_per_launch': int(round((2 * fetch + write) * 1024)),
      'algorithmic_bytes_per_launch': algorithmic,
  }, open(f'{out}/dreamer_pmc_gather.json', 'w'), indent=1)
except Exception as e:
  print('dreamer pmc summary skipped:', e)
PY
echo "wrote $O"
