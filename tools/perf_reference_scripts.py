"""The reference's own manual perf scripts, with their parameters, on the product
(GPU box) and on the numpy oracle beside it (host CPU, bounded counts):

  embodied/perf/test_replay.py:31-47   test_speed       length 32, capacity 1e5, chunksize 1024,
                                                        8 workers, STEP = 64x64x3 u8 image + 1024 f32
                                                        + 12 f32 + 3 flags; inserts/s, samples/s (batch 1)
  embodied/perf/test_replay.py:49-66   test_chunk_size  length 64, no capacity, chunksize 64 ... 4096
  embodied/perf/test_replay.py:68-76   test_removal     one worker, 1e5 items: evictions on every insert
  embodied/perf/test_driver.py:13-25   throughput_dummy Driver over 32 Dummy('disc') envs, RandomAgent

The reference's scripts call `replay.dataset(1)` (commented out upstream,
SURVEY.md 4); `sample(1)` is the same draw.  The product is driven exactly as
the scripts drive the reference: one `add(step, worker)` per host step dict --
the per-step compatibility path, not the vectorised device path `bench.py`
measures -- so these numbers say what a caller gets WITHOUT changing a line.
The `add_batch` line adds the same steps as one call per round of 8 workers
(host arrays) for comparison.  The oracle is TEST INFRASTRUCTURE, timed here as the CPU
baseline of this report only.

    python tools/perf_reference_scripts.py [--seconds 3] > profiles/rNN_perf_reference_scripts.txt
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STEP = {
    'image': np.zeros((64, 64, 3), np.uint8),
    'vector': np.zeros(1024, np.float32),
    'action': np.zeros(12, np.float32),
    'is_first': np.array(False),
    'is_last': np.array(False),
    'is_terminal': np.array(False),
}


# BASELINE.md 2 (the survey's indicative reference timings): the benchmark's own step
BASELINE_STEP = {
    'image': np.zeros((84, 84, 4), np.uint8),
    'reward': np.float32(0),
    'is_first': np.array(False),
    'is_last': np.array(False),
    'is_terminal': np.array(False),
    'action': np.int32(0),
}


def baseline_shape(make, label, seconds, device):
  """`Replay.add` (length 65, chunksize 1024, 64 workers, 84x84x4 u8 frames) and
  `sample(16)` -> (16, 65, 84, 84, 4): BASELINE.md 2's two replay rows."""
  global STEP
  keep, STEP = STEP, BASELINE_STEP
  try:
    replay = make(length=65, capacity=int(1e5), chunksize=1024)
    n, dt = timed_inserts(replay, int(3e5), 64, seconds, device)
  finally:
    STEP = keep
  done, start = 0, time.perf_counter()
  while time.perf_counter() - start < seconds:
    replay.sample(16)
    done += 1
  sync(device)
  ds = time.perf_counter() - start
  print(f'{label:<34} add steps/sec {n / dt:>9.0f} ({n} in {dt:.2f} s)   sample(16) batches/sec {done / ds:>8.0f} '
        f'= {done * 16 * 65 / ds:>9.0f} steps/sec ({len(replay)} items)', flush=True)


def sync(device):
  if device:
    import torch
    torch.cuda.synchronize()


def timed_inserts(replay, inserts, workers, seconds, device):
  """`inserts` adds round-robin over `workers`, cut off after `seconds`."""
  done, start = 0, time.perf_counter()
  per_round = workers
  while done < inserts:
    for worker in range(workers):
      replay.add(STEP, worker)
    done += per_round
    if done % (64 * per_round) == 0 and time.perf_counter() - start > seconds:
      break
  sync(device)
  return done, time.perf_counter() - start


def timed_batched_inserts(replay, inserts, workers, seconds, device):
  """The same steps, one `add_batch` per round of `workers` (host arrays)."""
  batch = {k: np.stack([v] * workers) for k, v in STEP.items()}
  ids = list(range(workers))
  done, start = 0, time.perf_counter()
  while done < inserts:
    replay.add_batch(batch, ids)
    done += workers
    if done % (64 * workers) == 0 and time.perf_counter() - start > seconds:
      break
  sync(device)
  return done, time.perf_counter() - start


def timed_samples(replay, samples, seconds, device):
  done, start = 0, time.perf_counter()
  while done < samples:
    replay.sample(1)
    done += 1
    if done % 256 == 0 and time.perf_counter() - start > seconds:
      break
  sync(device)
  return done, time.perf_counter() - start


def speed(make, label, seconds, device, length=32, capacity=1e5, chunksize=1024, workers=8,
          inserts=2e5, samples=1e5):
  replay = make(length=length, capacity=capacity and int(capacity), chunksize=chunksize)
  n, dt = timed_inserts(replay, int(inserts), workers, seconds, device)
  m, ds = timed_samples(replay, int(samples), seconds, device)
  print(f'{label:<34} inserts/sec {n / dt:>10.0f} ({n} in {dt:.2f} s)   '
        f'samples/sec {m / ds:>9.0f} ({m} in {ds:.2f} s)   items {len(replay)}', flush=True)


def removal(make, label, seconds, device):
  replay = make(length=32, capacity=int(1e5), chunksize=1024)
  # fill first (untimed beyond the cut-off), then time inserts that each evict
  n0, _ = timed_inserts(replay, int(1e5) + 31, 1, 120.0, device)
  n, dt = timed_inserts(replay, int(1e6), 1, seconds, device)
  print(f'{label:<34} inserts/sec {n / dt:>10.0f} with an eviction each ({n} in {dt:.2f} s, '
        f'{len(replay)} items held after {n0} to fill)', flush=True)


def driver_dummy(emb, label, seconds, parallel, envs=32, sink=False):
  from functools import partial as bind
  from embodied_amd.envs import dummy
  fns = [bind(dummy.Dummy, 'disc') for _ in range(envs)]
  example = fns[0]()
  agent = emb.RandomAgent(example.obs_space, example.act_space)
  example.close()
  driver = emb.Driver(fns, parallel)
  if sink:      # run/train.py:56-61: every transition goes into the replay
    replay = emb.Replay(length=32, capacity=int(1e5), chunksize=1024, seed=0)
    driver.on_step(replay.add)
  driver.reset(agent.init_policy)
  driver(agent.policy, steps=100)
  # The script credits every call with 100 * envs steps (test_driver.py:24); a
  # call of `steps=100` runs ceil(100 / envs) vectorised steps (driver.py:55-59,
  # 84: the count advances by the number of envs), so both figures are printed.
  calls, start = 0, time.perf_counter()
  while time.perf_counter() - start < seconds:
    driver(agent.policy, steps=100)
    calls += 1
  dt = time.perf_counter() - start
  driver.close()
  true_steps = calls * -(-100 // envs) * envs
  print(f'{label:<34} FPS as the script counts {calls * 100 * envs / dt:>10.0f}   env steps/sec '
        f'{true_steps / dt:>9.0f} ({envs} Dummy envs, parallel={parallel})', flush=True)


def oracle_driver_dummy(np_oracle, label, seconds, envs=32, sink=False):
  """The same loop over the oracle's serial Driver (driver.py:11-87), with the
  package's Dummy env and RandomAgent classes (host Python either way)."""
  from embodied_amd.envs import dummy
  from embodied_amd.core.agents import RandomAgent
  made = [dummy.Dummy('disc') for _ in range(envs)]
  agent = RandomAgent(made[0].obs_space, made[0].act_space)
  driver = np_oracle.Driver(made)
  if sink:
    replay = np_oracle.Replay(length=32, capacity=int(1e5), chunksize=1024, selector=np_oracle.Uniform(0))
    driver.on_step(replay.add)
  driver.reset(agent.init_policy)
  driver(agent.policy, steps=100)
  calls, start = 0, time.perf_counter()
  while time.perf_counter() - start < seconds:
    driver(agent.policy, steps=100)
    calls += 1
  dt = time.perf_counter() - start
  true_steps = calls * -(-100 // envs) * envs
  print(f'{label:<34} FPS as the script counts {calls * 100 * envs / dt:>10.0f}   env steps/sec '
        f'{true_steps / dt:>9.0f} ({envs} Dummy envs, serial)', flush=True)


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--seconds', type=float, default=3.0, help='cut-off per timed loop')
  p.add_argument('--no-oracle', action='store_true')
  p.add_argument('--no-product', action='store_true')
  args = p.parse_args()
  print(f'# tools/perf_reference_scripts.py --seconds {args.seconds}: the loops of embodied/perf/test_replay.py '
        f'and test_driver.py, each cut off after {args.seconds} s', flush=True)
  if not args.no_product:
    import torch
    import embodied_amd as emb
    print(f'# product: embodied_amd on {torch.cuda.get_device_name(0)}, host modules '
          f'{"compiled" if emb.compiled else "plain"}; one add(step, worker) per host step dict', flush=True)
    make = lambda **kw: emb.Replay(seed=0, **kw)
    # Once per process, outside the timed loops: the first launches load the
    # library's code objects (~0.2 s), which the first loop would otherwise carry.
    warm = make(length=2, capacity=64, chunksize=8)
    for t in range(40):
      warm.add(STEP, 0)
    warm.sample(1)
    sync(True)
    speed(make, 'product test_speed', args.seconds, True)
    replay = make(length=32, capacity=int(1e5), chunksize=1024)
    n, dt = timed_batched_inserts(replay, int(2e5), 8, args.seconds, True)
    print(f'{"product test_speed, add_batch of 8":<34} inserts/sec {n / dt:>10.0f} ({n} in {dt:.2f} s)', flush=True)
    for chunksize in (64, 128, 256, 512, 1024, 2048, 4096):
      speed(make, f'product test_chunk_size {chunksize}', args.seconds, True, length=64, capacity=None,
            chunksize=chunksize)
    removal(make, 'product test_removal', args.seconds, True)
    baseline_shape(make, 'product BASELINE.md 2 shapes', args.seconds, True)
    driver_dummy(emb, 'product throughput_dummy', args.seconds, False)
    driver_dummy(emb, 'product throughput_dummy', args.seconds, True)
    driver_dummy(emb, 'product dummy + on_step(add)', args.seconds, False, sink=True)
    driver_dummy(emb, 'product dummy + on_step(add)', args.seconds, True, sink=True)
  if not args.no_oracle:
    from oracle import np_oracle
    print(f'# cpu baseline: oracle/np_oracle.py (numpy restatement of the reference), 1 core of '
          f'{os.cpu_count()}', flush=True)
    make = lambda **kw: np_oracle.Replay(selector=np_oracle.Uniform(0), **kw)
    speed(make, 'oracle test_speed', args.seconds, False)
    for chunksize in (64, 1024, 4096):
      speed(make, f'oracle test_chunk_size {chunksize}', args.seconds, False, length=64, capacity=None,
            chunksize=chunksize)
    removal(make, 'oracle test_removal', args.seconds, False)
    baseline_shape(make, 'oracle BASELINE.md 2 shapes', args.seconds, False)
    oracle_driver_dummy(np_oracle, 'oracle throughput_dummy', args.seconds)
    oracle_driver_dummy(np_oracle, 'oracle dummy + on_step(add)', args.seconds, sink=True)


if __name__ == '__main__':
  main()
