"""Random Replay histories under the Prioritized selector, host index of the
library against the oracle (which `sel_prioritized` / `replay_prioritized` pin
to the reference): tests/test_host_index.py's fuzz with many more seeds and a
bias towards what ppo/configs.yaml:42 ships (initial inf, zero_on_sample) --
the settings whose refresh runs as one sliding pass (selectors.h refresh_drawn).
CPU only.   python tools/fuzz_prioritized.py --seeds 2000 [--first 0]"""
import argparse
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_amd import selectors                       # noqa: E402
from oracle import np_oracle                              # noqa: E402
from tests.conftest import assert_same                    # noqa: E402
from tests.test_host_index import HostReplay              # noqa: E402


def one(seed, steps):
  gen = np.random.default_rng([4242, seed])
  shipped = gen.random() < 0.6
  length = int(gen.integers(1, 12))
  chunksize = int(gen.integers(2, 16))
  capacity = int(gen.integers(4, 90))
  workers = int(gen.integers(1, 5))
  kw = dict(
      exponent=float(gen.choice([1.0, 0.8, 0.5])),
      maxfrac=float(gen.choice([0.0, 0.5, 1.0]) if not shipped else gen.choice([0.0, 0.5])),
      initial=float(np.inf if shipped else gen.choice([1.0, np.inf, 0.3])),
      zero_on_sample=bool(True if shipped else gen.integers(0, 2)),
      branching=int(gen.choice([2, 3, 16])), seed=seed)
  ours = HostReplay(length, capacity, chunksize, False, selector=selectors.Prioritized(**kw), n_slots=256)
  ref = np_oracle.Replay(length, capacity, chunksize, False, selector=np_oracle.Prioritized(**kw))
  clock, kept, drawn = [0] * workers, [], 0
  write_back = float(gen.choice([0.0, 0.3, 0.7]))        # 0: the priorities stay inf / zeroed (the plain case)
  for n in range(steps):
    w = int(gen.integers(0, workers))
    step = {'t': np.int32(clock[w]), 'w': np.int32(w)}
    clock[w] += 1
    ours.add(step, w)
    ref.add(step, w)
    assert len(ours) == len(ref), (seed, n)
    if len(ref) and n % int(gen.integers(1, 6)) == 0:
      batch = int(gen.integers(1, 6))
      try:
        want = ref.sample(batch)
      except ValueError as e:
        assert 'NaN' in str(e), e
        try:
          ours.sample(batch)
        except ValueError as e2:
          assert 'NaN' in str(e2), e2
          return drawn, 'nan'
        raise AssertionError(f'seed {seed} n {n}: the oracle refused NaN masses, the library did not')
      got = ours.sample(batch)
      assert_same(got, want, f'seed{seed} n{n} {kw}')
      drawn += batch
      kept.append(want['stepid'])
      if gen.random() < write_back:
        stepid = kept[int(gen.integers(0, len(kept)))]
        kind = gen.choice(['unit', 'ten', 'zero', 'inf'], p=[0.4, 0.3, 0.2, 0.1])
        prio = gen.random(stepid.shape[:2]) * {'unit': 1.0, 'ten': 10.0, 'zero': 0.0, 'inf': 1.0}[kind]
        if kind == 'inf':
          prio[gen.random(prio.shape) < 0.3] = np.inf
        ours.update({'stepid': stepid, 'priority': prio})
        ref.update({'stepid': stepid, 'priority': prio})
  return drawn, 'ok'


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--seeds', type=int, default=200)
  p.add_argument('--first', type=int, default=0)
  p.add_argument('--steps', type=int, default=400)
  args = p.parse_args()
  warnings.filterwarnings('ignore', message='invalid value encountered')
  total = nan = 0
  for seed in range(args.first, args.first + args.seeds):
    drawn, how = one(seed, args.steps)
    total += drawn
    nan += how == 'nan'
  print(f'fuzz_prioritized: seeds {args.first}..{args.first + args.seeds - 1} x {args.steps} steps: '
        f'{total} sampled sequences compared, {nan} histories ended in the NaN refusal both sides share, '
        f'no mismatch')


if __name__ == '__main__':
  main()
