"""`python bench.py --gpus N` must start N ranks itself (the driver runs exactly
that command form for N = 1, 2, 4, 8).  Two ranks on the test box's one GPU
with the gloo backend: the control flow is the multi-rank one, only the
transport differs from RCCL."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def fake_rccl():
  """The suite's loopback stand-in for RCCL (RCCL itself refuses two ranks on one
  GPU), bound behind the library's dlsym table with EMB_RCCL_LIB."""
  import importlib.util
  spec = importlib.util.spec_from_file_location('_fake_rccl_build', ROOT / 'tests' / 'fake_rccl' / 'build.py')
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return str(mod.build())


def run_bench(*flags, env=None):
  full = dict(os.environ, **(env or {}))
  # (HIP_FORCE_DEV_KERNARG: conftest pins the suite to the runtime's placement; the
  # bench is run as the driver runs it, with its own default)
  for name in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
               'HIP_FORCE_DEV_KERNARG'):
    full.pop(name, None)
  res = subprocess.run([sys.executable, str(ROOT / 'bench.py'), *flags], cwd=ROOT, env=full,
                       capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stderr[-3000:]
  lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, res.stdout[-2000:]        # ONE JSON line, from rank 0
  return json.loads(lines[0])


def test_bench_gpus_2_spawns_two_ranks():
  rec = run_bench('--gpus', '2', '--steps', '20', '--warmup', '5', '--backend', 'gloo')
  assert rec['n_gpus'] == 2 and rec['rccl_ranks'] == 2 and rec['backend'] == 'gloo'
  assert rec['steps'] == 20 and rec['warmup'] == 5
  assert rec['config']['global_envs'] == 2 * rec['config']['envs_per_gpu']
  assert rec['scaling'] == 'weak' and rec['value'] > 0
  assert rec['sustained']['seconds'] >= 2.0
  assert rec['roofline']['launches'] >= 1          # rank 0's gathers inside the timed region
  # context for reading a scaling curve: the same loop as N independent replicas
  assert rec['replicas_only']['env_steps_per_s'] > 0 and rec['replicas_only']['steps'] % 256 == 0


def test_bench_gpus_2_native_exchange_between_two_real_ranks():
  """`--comm auto` with two ranks: the run-time self-check of the library's own
  collective entry points (emb_comm_*) against torch.distributed passes between
  two REAL ranks and the timed path then issues one emb_comm_exchange per train
  step.  Transport: the suite's loopback stand-in for RCCL (RCCL itself refuses
  two ranks on one GPU), selected with EMB_RCCL_LIB; process group: gloo."""
  lib = fake_rccl()
  rec = run_bench('--gpus', '2', '--steps', '40', '--warmup', '5', '--capacity', '20000',
                  '--grad-numel', '100000', '--sustained-seconds', '1', '--prewarm-train-steps', '20',
                  '--backend', 'gloo', '--comm', 'native', env={'EMB_RCCL_LIB': lib})
  assert rec['n_gpus'] == 2 and rec['backend'] == 'gloo'
  native = rec['native_comm']
  assert native['status'] == 'ok' and native['ranks'] == 2 and all(native['checks'].values()), native
  assert native['transport'] == lib
  assert native['timed_path'] == 'native' and 'emb_comm_exchange' in rec['config']['parallelism']
  assert rec['train_steps_per_s'] > 0 and rec['value'] > 0
  exp = rec['expected']                         # DESIGN.md 5's link budget, printed with the line
  assert exp['replicas_only_x'] == 2.0 and exp['links'] == 1
  assert exp['link_time_us_at_60pct'] > exp['link_time_us_at_100pct'] > 0
  assert 0 < exp['link_bound_x_at_60pct'] <= 2.0 and exp['measured_x']['value'] > 0


def test_bench_gpus_2_direct_schedule_between_two_real_ranks():
  """`--comm direct`: the timed path's collectives go through emb_direct_exchange --
  every rank writing its peer's shares through hipIpc pointers (two processes on
  the test box's one GPU) -- after the self-check against torch.distributed
  passed; the line carries both transports' per-call times."""
  lib = fake_rccl()
  rec = run_bench('--gpus', '2', '--steps', '40', '--warmup', '5', '--capacity', '20000',
                  '--grad-numel', '100000', '--sustained-seconds', '1', '--prewarm-train-steps', '20',
                  '--backend', 'gloo', '--comm', 'direct', env={'EMB_RCCL_LIB': lib})
  native = rec['native_comm']
  direct = native['direct']
  assert direct['status'] == 'ok' and all(direct['checks'].values()) and direct['timed_out'] is False, direct
  assert set(direct['per_call']) == {'direct_all_reduce', 'direct_all_to_all', 'direct_all_gather',
                                     'direct_exchange_step', 'direct_exchange_gather_step'}
  assert set(direct['checks']) == {'all_to_all', 'all_gather', 'all_reduce_sum_f32', 'all_reduce_mean_f32',
                                   'all_reduce_sum_bf16', 'all_reduce_mean_bf16', 'exchange_gather'}
  assert native['timed_path'] == 'direct' and 'emb_direct_exchange' in rec['config']['parallelism']
  assert rec['train_steps_per_s'] > 0 and rec['value'] > 0
  assert native['per_train_step']['direct_collectives_us'] > 0
  assert set(rec['expected']['link_bound_x_measured']) == {'rccl', 'direct', 'c10d'}     # every transport, side by side
  assert rec['transports']['timed_path'] == 'direct' and rec['transports']['grad_dtype'] == 'f32'
  assert set(rec['transports']['exchange_step_us']) == {'c10d', 'rccl', 'direct'}
  assert rec['transports']['schedule_agreed_at_fences'] >= 2
  assert 0 < rec['expected']['link_bound_x_measured']['direct'] <= 2.0


def test_bench_auto_takes_the_faster_transport_that_passed_its_check():
  """`--comm auto` with two ranks: both transports are checked and timed on the
  job's own bytes, every rank takes the same choice (MAX over ranks of the
  measured exchange times).  Here the RCCL stand-in is a host-staged loopback,
  far slower than stores through hipIpc pointers on one GPU: auto takes direct."""
  rec = run_bench('--gpus', '2', '--steps', '40', '--warmup', '5', '--capacity', '20000',
                  '--grad-numel', '100000', '--sustained-seconds', '1', '--prewarm-train-steps', '20',
                  '--backend', 'gloo', env={'EMB_RCCL_LIB': fake_rccl()})
  native = rec['native_comm']
  assert native['status'] == 'ok' and native['direct']['status'] == 'ok'
  auto = native['auto']
  assert auto['chose'] == ('direct' if auto['direct_exchange_us'] < 0.9 * auto['rccl_exchange_us'] else 'native')
  assert native['timed_path'] == auto['chose'] and rec['value'] > 0


@pytest.mark.parametrize('fault,left', [
    ('direct_open:1', 'native'),      # rank 1 cannot make / map its hipIpc buffer
    ('direct_stuck:1', 'native'),     # rank 1 never arrives at one direct collective: rank 0's wait gives up
    ('direct_wrong:0', 'native'),     # rank 0's direct all-to-all delivers wrong bytes
    ('native_open:1', 'direct'),      # rank 1's RCCL communicator cannot be set up: the direct schedule is left
])
def test_bench_auto_degrades_when_a_transport_fails_on_one_rank(fault, left):
  """First contact with a node: a transport that fails its self-check on ONE rank
  -- set-up error, a peer that never arrives (the bounded wait gives up, the
  communicator dies on every rank), wrong bytes -- must cost the job seconds and
  nothing else: every rank takes the same decision, `--comm auto` goes on with
  what is left, the line is valid and the exit code 0."""
  rec = run_bench('--gpus', '2', '--steps', '40', '--warmup', '5', '--capacity', '20000',
                  '--grad-numel', '100000', '--sustained-seconds', '1', '--prewarm-train-steps', '20',
                  '--backend', 'gloo', env={'EMB_RCCL_LIB': fake_rccl(), 'EMB_BENCH_FAULT': fault})
  native = rec['native_comm']
  broken = native['direct'] if fault.startswith('direct') else native
  assert broken['status'] != 'ok', broken
  assert native['auto']['chose'] == left and native['timed_path'] == left, native['auto']
  assert ('emb_direct' if fault.startswith('direct') else 'emb_comm') in native['auto']['why']
  assert rec['transports']['timed_path'] == left
  assert rec['n_gpus'] == 2 and rec['value'] > 0 and rec['train_steps_per_s'] > 0
  assert rec['replicas_only']['env_steps_per_s'] > 0 and rec['expected']['links'] == 1
  assert rec['transports']['schedule_agreed_at_fences'] >= 2


def test_bench_auto_falls_back_to_c10d_when_both_transports_fail():
  """Neither transport passes (here: RCCL set-up fails on rank 1 AND a direct
  peer never arrives): the timed path stays on torch.distributed, rc 0."""
  # (one fault per run: the second failure is a transport library that does not load at all)
  rec = run_bench('--gpus', '2', '--steps', '40', '--warmup', '5', '--capacity', '20000',
                  '--grad-numel', '100000', '--sustained-seconds', '1', '--prewarm-train-steps', '20',
                  '--backend', 'gloo', env={'EMB_RCCL_LIB': '/nonexistent/librccl.so', 'EMB_BENCH_FAULT': 'direct_stuck:0'})
  native = rec['native_comm']
  assert native['status'] != 'ok' and native['direct']['status'] != 'ok'
  assert native['auto']['chose'] == 'c10d' and native['timed_path'] == 'c10d'
  assert 'torch.distributed' in rec['config']['parallelism']
  assert rec['value'] > 0 and rec['train_steps_per_s'] > 0


@pytest.mark.parametrize('comm,exchange', [
    ('direct', 'trajectories'), ('direct', 'online'), ('direct', 'returns'), ('native', 'trajectories')])
def test_bench_all_gather_forms_run_on_the_librarys_transports(comm, exchange):
  """north_star's "all-gather of trajectories": `--exchange trajectories | online |
  returns` go through the same transport as the gradients, ONE exchange call per
  train step (emb_direct_exchange_gather / emb_comm_exchange_gather)."""
  rec = run_bench('--gpus', '2', '--steps', '40', '--warmup', '5', '--capacity', '20000',
                  '--grad-numel', '100000', '--sustained-seconds', '1', '--prewarm-train-steps', '20',
                  '--backend', 'gloo', '--comm', comm, '--exchange', exchange, env={'EMB_RCCL_LIB': fake_rccl()})
  native = rec['native_comm']
  assert native['timed_path'] == comm and rec['transports']['exchange'] == exchange
  step = native['per_train_step']
  assert step['trajectory_collective'] == 'all_gather'
  assert step['trajectory_share'] > 0 and (exchange == 'online' or step['trajectory_share'] == 1.0)
  assert set(rec['transports']['exchange_step_us']) == {'c10d', 'rccl', 'direct'}
  assert rec['expected']['bytes_one_way_per_train_step'] > 2 * 0.5 * 400000      # gradients + the gathered blocks
  assert rec['value'] > 0 and rec['train_steps_per_s'] > 0


def test_bench_gpus_8_control_flow():
  """The driver's largest form, `python bench.py --gpus 8`: eight ranks start,
  agree on their exchange schedule at every fence and rank 0 prints one line
  (gloo: the eight ranks share the test box's one GPU)."""
  rec = run_bench('--gpus', '8', '--steps', '20', '--warmup', '5', '--capacity', '6000',
                  '--grad-numel', '100000', '--sustained-seconds', '1', '--prewarm-train-steps', '10', '--backend', 'gloo')
  assert rec['n_gpus'] == 8 and rec['rccl_ranks'] == 8 and rec['backend'] == 'gloo'
  assert rec['config']['global_envs'] == 8 * rec['config']['envs_per_gpu']
  assert rec['scaling'] == 'weak' and rec['value'] > 0 and rec['train_steps_per_s'] > 0
  assert rec['replicas_only']['env_steps_per_s'] > 0
  assert rec['expected']['replicas_only_x'] == 8.0 and rec['expected']['links'] == 7


def test_bench_gpus_8_rehearsal_on_one_gpu():
  """`python bench.py --gpus 8` as the driver will run it on an 8-GPU node, rehearsed on
  the test box's ONE GPU (gloo process group, eight ranks sharing the device; the direct
  schedule's hipIpc stores stay on it): launcher, both transports' self-checks on the
  job's own bytes, the transport choice, 800 steps with 150 exchanges,
  the exchange schedule compared at every fence, the sustained window, the replicas-only
  leg, every field of the line.  Everything but the links is the real thing.

  What one GPU cannot rehearse is the default gradient (49 MB of f32 per train step):
  eight processes time-share the device, a rank's kernel spins on flags while its peers'
  kernels wait for the GPU, and ONE such all-reduce takes ~100 ms (profiles/
  r06_rehearsal_gpus8_one_gpu.json holds a run with the default size).  The gradient here
  is 400 KB; its dtype is the default f32."""
  rec = run_bench('--gpus', '8', '--backend', 'gloo', '--no-cpu-baseline', '--steps', '800', '--warmup', '50',
                  '--sustained-seconds', '1', '--grad-numel', '100000', '--capacity', '20000', '--prewarm-train-steps', '20',
                  env={'EMB_RCCL_LIB': fake_rccl()})
  assert rec['n_gpus'] == 8 and rec['rccl_ranks'] == 8 and rec['backend'] == 'gloo'
  assert rec['steps'] == 800 and rec['warmup'] == 50 and rec['sustained']['seconds'] >= 1
  assert rec['config']['global_envs'] == 8 * 64 and rec['scaling'] == 'weak'
  assert 'f32 grad all-reduce' in rec['config']['parallelism'] and 'LOWER precision' not in rec['config']['parallelism']
  t = rec['transports']
  assert t['grad_dtype'] == 'f32' and t['grad_bytes'] == 400000 and t['exchange'] == 'dp_slice'
  assert set(t['exchange_step_us']) == {'c10d', 'rccl', 'direct'}
  assert t['schedule_agreed_at_fences'] >= 5             # before / after the region, the sustained window, replicas-only
  native = rec['native_comm']
  assert native['status'] == 'ok' and native['direct']['status'] == 'ok', native
  assert native['auto']['chose'] == t['timed_path'] and native.get('direct_timed_out_during_run') in (None, False)
  exp = rec['expected']
  assert exp['links'] == 7 and exp['replicas_only_x'] == 8.0
  assert set(exp['link_bound_x_measured']) == {'rccl', 'direct', 'c10d'}
  assert rec['value'] > 0 and rec['train_steps_per_s'] > 0
  assert rec['regions']['train_steps'][0] >= 140         # well over a hundred exchanges inside the timed region
  # The N = 1 line of the same build: same metric, same workload, same per-rank configuration.
  # Eight processes time-sharing the one GPU (collectives off) deliver a fraction of what one
  # process does alone on it -- context switches between their queues -- so the rates are held
  # against each other with wide bounds only: the rehearsal checks agreement of the PROGRAM.
  single = run_bench('--no-cpu-baseline', '--no-context', '--no-dreamer-leg', '--sustained-seconds', '1',
                     '--steps', '800', '--warmup', '50', '--capacity', '20000')
  assert single['metric'] == rec['metric'] and single['unit'] == rec['unit']
  assert single['config']['workload'] == rec['config']['workload']
  assert single['config']['env_actions']['value_measured_with'] == rec['config']['env_actions']['value_measured_with']
  ratio = rec['replicas_only']['env_steps_per_s'] / single['sustained']['env_steps_per_s']
  assert 0.02 < ratio < 2.5, (ratio, rec['replicas_only'], single['sustained'])


def test_bench_dreamer_workload_with_ranks():
  """configs[3]'s shape of parallelism: per-rank Replay (sample, lambda-return,
  latent write-back) and one gradient all-reduce per train step."""
  rec = run_bench('--gpus', '2', '--workload', 'dreamer', '--capacity', '20000', '--steps', '40', '--grad-numel', '100000',
                  '--warmup', '5', '--sustained-seconds', '1', '--prewarm-train-steps', '20', '--backend', 'gloo')
  assert rec['n_gpus'] == 2 and rec['rccl_ranks'] == 2
  assert 'per-rank Replay' in rec['config']['parallelism']
  assert 'lambda-return' in rec['config']['workload']
  assert rec['train_steps_per_s'] > 0 and rec['roofline']['launches'] >= 1


def test_bench_checks_the_native_collectives_when_it_runs_on_rccl():
  """With RCCL as the transport the line carries `native_comm`: the emb_comm_*
  entry points against torch.distributed on the same GPUs (one rank here)."""
  rec = run_bench('--steps', '200', '--warmup', '50', '--sustained-seconds', '0', '--no-cpu-baseline',
                  '--no-context', '--force-dist')
  assert rec['backend'] == 'nccl' and rec['rccl_ranks'] == 1
  native = rec['native_comm']
  assert native['status'] == 'ok' and all(native['checks'].values()), native
  assert set(native['checks']) == {
      'all_gather', 'all_to_all', 'all_reduce_sum_f32', 'all_reduce_mean_f32',
      'all_reduce_sum_bf16', 'all_reduce_mean_bf16', 'exchange_gather'}
  assert native['per_call']['native_all_reduce']['host_us'] > 0
  # ... and, having passed, they carry the timed path (one emb_comm_exchange per train step).
  assert native['timed_path'] == 'native' and 'emb_comm_exchange' in rec['config']['parallelism']
  assert native['per_train_step']['collectives_us'] > 0 and native['per_train_step']['issue_period_us'] > 0
  assert rec['train_steps_per_s'] > 0 and rec['roofline']['launches'] >= 1


def test_bench_comm_c10d_keeps_torch_distributed_in_the_timed_path():
  rec = run_bench('--steps', '200', '--warmup', '50', '--sustained-seconds', '0', '--no-cpu-baseline',
                  '--no-context', '--comm', 'c10d', '--force-dist')
  assert 'native_comm' not in rec and 'torch.distributed' in rec['config']['parallelism']
  rec = run_bench('--steps', '200', '--warmup', '50', '--sustained-seconds', '0', '--no-cpu-baseline',
                  '--no-context', '--workload', 'dreamer', '--capacity', '20000', '--force-dist')
  assert rec['native_comm']['timed_path'] == 'native' and rec['train_steps_per_s'] > 0


def test_bench_short_run_keeps_its_shape():
  """The driver's short form on one GPU: the line carries roofline, the
  sustained window and the CPU baseline; the headline region is exactly --steps."""
  rec = run_bench('--steps', '20', '--warmup', '5', '--cpu-seconds', '2', '--sustained-seconds', '2')
  assert rec['n_gpus'] == 1 and rec['steps'] == 20
  roof = rec['roofline']
  assert roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and 0 < roof['frac'] < 1
  assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
  assert set(roof['batches_per_launch_sweep']) == {'1', '4', '8', '16', '64', 'how'}
  assert rec['sustained']['seconds'] >= 2.0 and rec['sustained']['gather_launches'] > 100
  assert rec['fresh_batches']['env_steps_per_s'] > 0 and rec['fresh_batches']['train_steps_per_s'] > 0
  assert rec['cpu_baseline']['kind'] == 'port' and rec['cpu_baseline']['cores'] == 1
  assert rec['config']['kernargs'] in ('host', 'device')
  assert rec['config']['host_modules'] in ('compiled', 'python')
  cpus = rec['config']['cpus']                    # pinned to a few idle CPUs on a big shared host
  assert cpus is None or (len(cpus) == 4 and cpus == sorted(cpus))


def test_default_single_gpu_run_appends_the_dreamer_workload():
  """BASELINE configs[2] rides in the driver's own command: the PPO line carries
  `workloads.dreamer` with the rates and the gather / write-back roofline figures
  of a short DreamerV3-shaped run (1M-step uniform replay, latents written back)."""
  rec = run_bench('--steps', '20', '--warmup', '5', '--sustained-seconds', '0', '--no-cpu-baseline',
                  '--no-context', '--dreamer-leg-steps', '400')
  assert rec['config']['workload'].startswith('ppo_atari_pong_64env')     # metric / value unchanged
  assert 'span_move_kernel' in rec['roofline']['kernel'] or 'flat_move_kernel' in rec['roofline']['kernel']
  leg = rec['workloads']['dreamer']
  assert 'error' not in leg, leg
  assert 'dreamerv3_1M_uniform' in leg['workload'] and 'capacity=1000000' in leg['workload']
  assert leg['env_steps_per_s'] > 0 and leg['train_steps_per_s'] > 0
  # the replay-context latents as the shipped agent reads them: K = 1 step of them per sequence
  assert leg['gather']['bytes_per_launch'] == 2 * 16 * (65 * 28255 + 1 * 40960)
  assert 0 < leg['gather']['frac'] < 1 and leg['gather']['launches'] >= 1
  assert leg['writeback']['bytes_per_launch'] == 2 * 16 * 65 * 40960       # every step is written back
  assert 0 < leg['writeback']['frac'] < 1 and 'Replay.update' in leg['writeback']['kernel']
  full = leg['full_gather']                      # ... and all L steps of them, as rounds 1-4 measured
  assert 'error' not in full, full
  assert full['gather']['bytes_per_launch'] == 2 * 16 * 65 * (28255 + 40960)
  assert full['env_steps_per_s'] > 0 and 0 < full['gather']['frac'] < 1


def test_short_form_value_is_the_median_of_sixteen_regions():
  """`--steps 20` (the driver's form): sixteen fenced regions of exactly 20 steps,
  `value` = the median one; every region's time is in the line."""
  rec = run_bench('--steps', '20', '--warmup', '5', '--sustained-seconds', '0', '--no-cpu-baseline',
                  '--no-context', '--no-dreamer-leg')
  regions = rec['regions']
  assert regions['n'] == 16 and len(regions['ms']) == 16 and len(regions['train_steps']) == 16
  assert rec['steps'] == 20
  ms = sorted(regions['ms'])
  assert abs(rec['ms_per_step'] * 20 - ms[7]) < 1e-3           # the lower median, a measured region
  rates = regions['env_steps_per_s']
  assert rates['min'] <= rates['median'] <= rates['max'] and rates['median'] == rec['value']
  assert abs(rec['value'] - 20 * 64 / (ms[7] * 1e-3)) / rec['value'] < 1e-3
  long = run_bench('--steps', '300', '--warmup', '5', '--sustained-seconds', '0', '--no-cpu-baseline',
                   '--no-context', '--no-dreamer-leg')
  assert long['regions']['n'] == 1


def test_bench_refuses_more_rccl_ranks_than_gpus():
  import torch
  want = torch.cuda.device_count() + 1
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', )}
  res = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', str(want)], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=300)
  assert res.returncode != 0 and 'GPU' in (res.stderr + res.stdout)
