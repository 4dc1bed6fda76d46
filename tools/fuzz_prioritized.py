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


def _sid(n):
  return np.frombuffer(int(n).to_bytes(20, 'big'), np.uint8)


def protocol(seed, ops, others):
  """The SELECTOR PROTOCOL alone (selectors.py:128-197: __setitem__, __delitem__,
  prioritize, __call__), the library's Prioritized against `others` (name ->
  class: the oracle's, the real reference's): windows over a small universe of
  step ids -- sliding ones (the stream representation) and arbitrary ones (the
  general one) -- and priorities for ANY step id of the universe, also ids that
  belong to no item yet (the table keeps them, :143-150) or any more.  Returns
  the number of draws compared."""
  gen = np.random.default_rng([777, seed])
  length = int(gen.integers(2, 6))
  universe = int(gen.integers(length + 8, 70))
  kw = dict(
      exponent=float(gen.choice([1.0, 0.8, 0.5])), maxfrac=float(gen.choice([0.0, 0.5])),
      initial=float(gen.choice([1.0, np.inf, 0.3])), zero_on_sample=bool(gen.integers(0, 2)),
      branching=int(gen.choice([2, 3, 16])), seed=seed)
  sels = {'library': selectors.Prioritized(**kw), **{name: cls(**kw) for name, cls in others.items()}}
  sliding = gen.random() < 0.6             # items k = steps k..k+length-1, oldest deleted first (mostly)
  live, next_key, head, drawn = [], 0, 0, 0

  def every(fn):
    return {name: fn(sel) for name, sel in sels.items()}

  for n in range(ops):
    op = gen.choice(['insert', 'delete', 'prioritize', 'draw'], p=[0.35, 0.15, 0.2, 0.3])
    if op == 'insert':
      if sliding and head + length <= universe:
        steps = list(range(head, head + length))
        head += 1
      elif sliding:
        continue
      else:
        steps = sorted(gen.choice(universe, size=length, replace=False).tolist())
      ids = np.stack([_sid(x) for x in steps])
      for sel in sels.values():
        sel[next_key] = ids
      live.append(next_key)
      next_key += 1
    elif op == 'delete' and len(live) > 2:
      key = live.pop(0 if (sliding and gen.random() < 0.9) else int(gen.integers(0, len(live))))
      for sel in sels.values():
        del sel[key]
    elif op == 'prioritize':
      steps = gen.choice(universe, size=int(gen.integers(1, 8)), replace=False)
      prios = gen.random(len(steps)) * float(gen.choice([1.0, 10.0, 0.0]))
      if gen.random() < 0.15:
        prios[gen.random(len(prios)) < 0.4] = np.inf
      ids = np.stack([_sid(x) for x in steps])
      for sel in sels.values():
        sel.prioritize(ids, prios)
    elif op == 'draw' and live:
      got = {}
      for name, sel in sels.items():
        try:
          got[name] = int(sel())
        except ValueError as e:
          assert 'NaN' in str(e), e
          got[name] = 'nan'
      assert len(set(got.values())) == 1, (seed, n, kw, got)
      if got['library'] == 'nan':
        return drawn
      drawn += 1
    sizes = every(len)
    assert len(set(sizes.values())) == 1, (seed, n, sizes)
  return drawn


def reference_classes():
  """name -> Prioritized class: the oracle's and, where /root/reference is (the
  build container), the real reference's."""
  others = {'oracle': np_oracle.Prioritized}
  try:
    from tests import adapters
    others['reference'] = adapters.reference_ns().Prioritized
  except Exception:
    pass
  return others


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--seeds', type=int, default=200)
  p.add_argument('--first', type=int, default=0)
  p.add_argument('--steps', type=int, default=400)
  p.add_argument('--protocol', action='store_true',
                 help='the selector protocol alone (also priorities for step ids that belong to no item), '
                      'against the oracle and -- in the build container -- the real reference class')
  args = p.parse_args()
  warnings.filterwarnings('ignore', message='invalid value encountered')
  if args.protocol:
    others = reference_classes()
    total = sum(protocol(seed, args.steps, others) for seed in range(args.first, args.first + args.seeds))
    print(f'fuzz_prioritized --protocol: seeds {args.first}..{args.first + args.seeds - 1} x {args.steps} '
          f'operations, library vs {sorted(others)}: {total} draws compared, no mismatch')
    return
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
