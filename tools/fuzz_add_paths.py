"""Random histories through every way a step can enter a Replay, against the
numpy oracle (GPU box; the oracle is the checker).  Per seed: random length /
chunksize / capacity / workers / stage_rows / slots (small pools grow), every
step goes in by one of: `add` of exact-dtype host values (the C call), `add` of
awkward values (lists, float64, strided: the Python conversion), `add_batch` of
host arrays for a run of workers (staged n rows), `add_batch` of device tensors;
samples are compared bit for bit at random points.

    python tools/fuzz_add_paths.py [--seeds 300] [--steps 400]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import embodied_amd as emb
from oracle import np_oracle
from tests import scenarios
from tests.conftest import assert_same


def awkward(step, gen):
  out = dict(step)
  pick = int(gen.integers(0, 4))
  if pick == 0:
    out['vec'] = step['vec'].astype(np.float64)
  elif pick == 1:
    out['vec'] = step['vec'].tolist()
  elif pick == 2:
    out['image'] = np.repeat(step['image'], 2, axis=1)[:, ::2]
  else:
    out['reward'] = float(step['reward'])
    out['log/extra'] = 3.0
  return out


def one(seed, steps):
  gen = np.random.default_rng(1000 + seed)
  length = int(gen.integers(1, 9))
  chunksize = int(gen.integers(2, 14))
  capacity = None if gen.integers(0, 5) == 0 else int(gen.integers(2, 80))
  online = bool(gen.integers(0, 2))
  workers = int(gen.integers(1, 7))
  ours = emb.Replay(length, capacity, chunksize=chunksize, online=online, seed=seed,
                    stage_rows=int(gen.integers(1, 65)), slots=int(gen.integers(6, 12)))
  ref = np_oracle.Replay(length, capacity, chunksize, online, seed=seed)
  clock = [0] * workers
  ways = [0, 0, 0, 0]
  first = True
  n = 0
  while n < steps:
    way = 0 if first else int(gen.integers(0, 4))
    first = False
    if way in (0, 1):
      w = int(gen.integers(0, workers))
      step = scenarios.synth_step(clock[w], w)
      clock[w] += 1
      ours.add(step if way == 0 else awkward(step, gen), w)
      ref.add(step, w)
      n += 1
    else:
      k = int(gen.integers(1, workers + 1))
      ids = sorted(gen.choice(workers, size=k, replace=False).tolist())
      rows = []
      for w in ids:
        rows.append(scenarios.synth_step(clock[w], w))
        clock[w] += 1
        ref.add(rows[-1], w)
      batch = {key: np.stack([r[key] for r in rows]) for key in rows[0]}
      if way == 3:
        batch = {key: torch.from_numpy(value).cuda() for key, value in batch.items()}
      ours.add_batch(batch, ids)
      n += k
    ways[way] += 1
    assert len(ours) == len(ref), (seed, n)
    if len(ref) and gen.integers(0, 9) == 0:
      mode = ('train', 'report')[int(gen.integers(0, 2))]
      got = {k: v.cpu().numpy() for k, v in ours.sample(3, mode).items()}
      assert_same(got, ref.sample(3, mode), f'seed{seed} n{n}')
  got, want = ours.stats(), ref.stats()
  for k in ('items', 'chunks', 'streams', 'inserts', 'samples', 'updates'):
    assert got[k] == want[k], (seed, k, got[k], want[k])
  return ways


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--seeds', type=int, default=300)
  p.add_argument('--steps', type=int, default=400)
  args = p.parse_args()
  start, total = time.time(), [0, 0, 0, 0]
  for seed in range(args.seeds):
    ways = one(seed, args.steps)
    total = [a + b for a, b in zip(total, ways)]
  print(f'{args.seeds} seeds x {args.steps} steps: no mismatch against the oracle in {time.time() - start:.0f} s; '
        f'calls by way: add exact {total[0]}, add awkward {total[1]}, add_batch host {total[2]}, '
        f'add_batch device {total[3]}')


if __name__ == '__main__':
  main()
