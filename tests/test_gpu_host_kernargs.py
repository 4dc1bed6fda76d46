"""Process-wide settings that are read once, each in a child process.  The suite
itself runs on the package's defaults (kernel arguments in host memory, the
movers' indirect-argument kernels, index bookkeeping on the helper thread); the
children cover the other choices: the HIP runtime's own argument placement
(HIP_FORCE_DEV_KERNARG=1), the flat mover only, the argument-writer kernel, the
plain Python modules, bookkeeping on the calling thread, the helper thread's
default pace rule."""
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_parity_with_kernel_arguments_in_device_memory():
  """HIP_FORCE_DEV_KERNARG=1 (the runtime's default, read at HIP start-up): the
  movers take their by-value argument blocks, inserts carry completion stamps."""
  env = dict(os.environ, HIP_FORCE_DEV_KERNARG='1')
  res = subprocess.run(
      [sys.executable, '-m', 'pytest', 'tests/test_gpu_parity.py', 'tests/test_gpu_early_insert.py',
       '-m', 'gpu', '-q', '-x',
       '-k', '(golden or full_size or large_tables or update_roundtrip or sharded or very_large_rows '
             'or span_mover or fused_sample or early or many_envs or host_envs or checkpoint_between '
             'or wide_observations or parallel_env_workers) '          # (round 6: slab pieces by kernel, actions by store)
             'and not soak and not readers_on_other'],
      cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
  assert ' passed' in res.stdout


def test_parity_with_the_flat_mover_only():
  """EMB_SPAN_MOVER=0 switches the persistent span mover off (the knob is
  read once per process): sample, windowing, write-back and the grouped
  (per-destination-rank) layout must give the same bytes through the flat mover,
  which is what moves > 40 MB launches with device-resident arguments."""
  env = dict(os.environ, EMB_SPAN_MOVER='0')
  res = subprocess.run(
      [sys.executable, '-m', 'pytest', 'tests/test_gpu_parity.py', 'tests/test_gpu_heads.py', '-m', 'gpu', '-q', '-x',
       # (with host-resident arguments these all run the indirect flat movers, whose tables AND
       # spans are staged through LDS since round 6; heads: per-key lengths through that path)
       '-k', 'golden or full_size or span_mover or fused_sample or grouped or update_table '
             'or window or random_schemas or heads'],
      cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
  assert ' passed' in res.stdout


def test_parity_with_the_argument_writer_kernel():
  """Host-resident kernel arguments on a system WITHOUT a large BAR: the movers'
  argument blocks reach device memory through the one-workgroup writer kernel
  (EMB_ARGS_BAR=0 forces that route here)."""
  env = dict(os.environ, HIP_FORCE_DEV_KERNARG='0', EMB_ARGS_BAR='0')
  res = subprocess.run(
      [sys.executable, '-m', 'pytest', 'tests/test_gpu_parity.py', '-m', 'gpu', '-q', '-x',
       '-k', 'full_size or update_roundtrip or very_large_rows or fused_sample'],
      cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
  assert ' passed' in res.stdout


def test_parity_with_the_plain_python_modules():
  """Git tracks the `.py` sources; the GPU suite normally runs their Cython
  copies (embodied_amd/_compiled).  EMB_PURE_PYTHON=1 (read at import) runs the
  plain modules: the golden scenarios, the full-size sample, the device Driver,
  the early insert and the replay suite must give the same bytes.  The child also
  proves which form it ran."""
  env = dict(os.environ, EMB_PURE_PYTHON='1')
  probe = subprocess.run(
      [sys.executable, '-c',
       'import embodied_amd as e; print(sorted(e.compiled.loaded)); '
       'import embodied_amd.core.replay as r; print(r.__file__)'],
      cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
  assert probe.returncode == 0, probe.stderr[-2000:]
  assert probe.stdout.splitlines()[0] == '[]', probe.stdout      # no compiled module in use
  assert probe.stdout.splitlines()[1].endswith('embodied_amd/core/replay.py'), probe.stdout
  res = subprocess.run(
      [sys.executable, '-m', 'pytest', 'tests/test_gpu_parity.py', 'tests/test_gpu_early_insert.py',
       'tests/test_gpu_replay_suite.py', '-m', 'gpu', '-q', '-x',
       '-k', '(golden or full_size or device_driver or mask or early or replay or update_table '
             'or random_histories) and not soak and not readers_on_other'],
      cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
  assert ' passed' in res.stdout


def test_early_insert_with_the_index_bookkeeping_on_the_calling_thread():
  """EMB_DEFER_INDEX=0 (read once per process): emb_replay_publish does its index
  bookkeeping itself instead of posting it to the helper thread."""
  env = dict(os.environ, EMB_DEFER_INDEX='0')
  res = subprocess.run(
      [sys.executable, '-m', 'pytest', 'tests/test_gpu_early_insert.py', '-m', 'gpu', '-q', '-x',
       '-k', 'not soak and not predicted_rows'],
      cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
  assert ' passed' in res.stdout


def test_helper_thread_pace_rule_at_its_default():
  """Without EMB_DEFER_MAX_GAP_US (conftest lifts it for the suite) only a loop
  that publishes every < ~80 us hands its index bookkeeping to the helper
  thread: a tight stepping loop does, a loop that pauses 2 ms per step does
  not -- and both leave the replay the oracle expects (the early-insert parity
  case of the suite, at the default gate)."""
  env = {k: v for k, v in os.environ.items() if k != 'EMB_DEFER_MAX_GAP_US'}
  code = '''
import time, torch
import embodied_amd as emb
from embodied_amd.envs import synthetic
def run(pause, steps=300):
  n = 8
  env = synthetic.SyntheticBatchEnv(n, shape=(8, 8, 4), episode_len=50, ring=4)
  rep = emb.Replay(length=3, capacity=4000, chunksize=64, online=True, seed=0)
  drv = emb.Driver(batch_env=env, device='cuda')
  drv.on_step(rep.add)
  act = {'action': torch.zeros(n, dtype=torch.int32, device='cuda')}
  def policy(carry, obs, **kw):
    emb.ops.obs_stack(obs['image'], layout='channels_first', dtype=torch.bfloat16, scale=1 / 255)
    return carry, act, {}
  drv.reset()
  for _ in range(steps):
    drv(policy, steps=n)
    if pause:
      time.sleep(pause)
  torch.cuda.synchronize()
  return rep.early_inserts, rep.profile_report('deferred')[0], len(rep)
fast = run(0)
slow = run(0.002)
print('RESULT', fast, slow)
assert fast[0] >= 298 and slow[0] >= 298, (fast, slow)      # the early insert ran either way
assert fast[1] >= 250, fast                                 # tight loop: deferred
assert slow[1] <= 3, slow                                   # 2 ms between publishes: not deferred
assert fast[2] == slow[2], (fast, slow)
'''
  res = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
  assert 'RESULT' in res.stdout
