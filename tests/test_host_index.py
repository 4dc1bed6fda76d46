"""Host index core of libembodied_hip.so (selectors, SampleTree, ReplayIndex)
against the golden vectors recorded from the reference and against the oracle.
No GPU: payload bytes are kept in numpy by a test-side pool so that only the
C++ integer/float64 bookkeeping is under test here."""
import ctypes as C

import numpy as np
import pytest

from embodied_amd import _lib, selectors
from embodied_amd._lib import api
from oracle import np_oracle
from tests import adapters, scenarios
from tests.conftest import assert_same, load_golden


class HostReplay:
  """Drives emb_replay_*_index and mirrors rows into numpy (TEST stand-in for
  the device pool; the product's Replay moves rows with HIP kernels)."""

  def __init__(self, length, capacity=None, chunksize=1024, online=False,
               selector=None, seed=0, n_slots=256):
    self.length, self.chunksize, self.n_slots = length, chunksize, n_slots
    cfg = _lib.ReplayConfig(length, capacity or 0, chunksize, n_slots, int(online), 0, 0, 1, 0)
    self.selector = selector
    self.h = C.c_void_p()
    api.emb_replay_create(
        C.byref(cfg), selector._handle if selector is not None else None,
        seed, C.byref(self.h))
    self.pool = None

  def __del__(self):
    api.raw.emb_replay_destroy(self.h)

  def __len__(self):
    n = C.c_int64()
    api.emb_replay_len(self.h, C.byref(n))
    return n.value

  def add(self, step, worker=0):
    step = {k: np.asarray(v) for k, v in step.items() if not k.startswith('log/')}
    workers = np.array([worker], np.int64)
    rows = np.zeros(1, np.int32)
    sid = np.zeros((1, 20), np.uint8)
    api.emb_replay_add_index(self.h, 1, _lib.ptr(workers), _lib.ptr(rows), _lib.ptr(sid), None)
    step['stepid'] = sid[0]
    if self.pool is None:
      self.pool = {k: np.zeros((self.n_slots * self.chunksize, *v.shape), v.dtype)
                   for k, v in step.items()}
    for k, v in step.items():
      self.pool[k][rows[0]] = v

  def sample_rows(self, batch, mode='train'):
    rows = np.zeros((batch, self.length), np.int32)
    online = np.zeros(batch, np.uint8)
    api.emb_replay_sample_index(
        self.h, batch, _lib.MODES[mode], _lib.ptr(rows), _lib.ptr(online), None)
    return rows, online

  def sample(self, batch, mode='train'):
    rows, _ = self.sample_rows(batch, mode)
    return np_oracle.annotate({k: v[rows] for k, v in self.pool.items()})

  def update(self, data):
    data = dict(data)
    stepid = np.ascontiguousarray(data.pop('stepid'))
    priority = data.pop('priority', None)
    if priority is not None:
      flat = np.ascontiguousarray(stepid.reshape(-1, 20))
      prios = np.ascontiguousarray(priority, np.float64).reshape(-1)
      api.emb_replay_prioritize(self.h, _lib.ptr(flat), _lib.ptr(prios), len(prios))
    if not data:
      return
    B = len(stepid)
    T = len(next(iter(data.values()))[0])
    first = np.ascontiguousarray(stepid[:, 0])
    rows = np.zeros((B, T), np.int32)
    found = np.zeros(B, np.uint8)
    api.emb_replay_resolve(self.h, B, _lib.ptr(first), T, _lib.ptr(rows), _lib.ptr(found))
    for b in range(B):
      if found[b]:
        for k, v in data.items():
          self.pool[k][rows[b]] = np.asarray(v)[b]

  def stats(self):
    out = np.zeros(6, np.int64)
    api.emb_replay_stats(self.h, _lib.ptr(out), 1)
    names = ('items', 'chunks', 'streams', 'inserts', 'samples', 'updates')
    return dict(zip(names, out.tolist()))


def host_ns():
  ns = adapters.oracle_ns()
  ns.name = 'host-index'
  ns.Replay = HostReplay
  ns.Uniform = selectors.Uniform
  ns.Prioritized = selectors.Prioritized
  ns.Mixture = selectors.Mixture
  ns.SampleTree = selectors.SampleTree
  ns.consec = lambda rep, batch, length, consec, prefix: iter(np_oracle.Consec(
      lambda: rep.sample(batch, 'train'), length, consec, prefix))
  return ns


HOST_SCENARIOS = [n for n in sorted(scenarios.SCENARIOS) if n != 'driver_script']


@pytest.mark.parametrize('name', HOST_SCENARIOS)
def test_host_index_matches_reference_golden(name):
  got = scenarios.SCENARIOS[name](host_ns())
  assert_same(got, load_golden(name), name)


def test_foreign_selector_through_callbacks():
  """Any Python object with the selector protocol can drive the native index."""
  native = HostReplay(length=3, capacity=9, chunksize=4, seed=5)
  foreign = HostReplay(length=3, capacity=9, chunksize=4,
                       selector=selectors.Foreign(np_oracle.Uniform(5)))
  for t in range(40):
    for w in range(2):
      native.add({'t': np.int32(t)}, w)
      foreign.add({'t': np.int32(t)}, w)
    assert len(native) == len(foreign)
  a, _ = native.sample_rows(16)
  b, _ = foreign.sample_rows(16)
  assert (a == b).all()
  foreign.selector.reraise()


@pytest.mark.parametrize('seed', range(4))
def test_random_histories_against_oracle(seed):
  """Fuzz: random worker interleavings, capacities and chunk sizes; every
  sampled row table must address exactly the steps the oracle returns."""
  gen = np.random.default_rng(seed)
  length = int(gen.integers(1, 9))
  chunksize = int(gen.integers(1, 12))
  capacity = int(gen.integers(1, 40))
  online = bool(gen.integers(0, 2))
  workers = int(gen.integers(1, 6))
  ours = HostReplay(length, capacity, chunksize, online, seed=seed, n_slots=256)
  ref = np_oracle.Replay(length, capacity, chunksize, online, seed=seed)
  clock = [0] * workers
  for n in range(600):
    w = int(gen.integers(0, workers))
    step = {'t': np.int32(clock[w]), 'w': np.int32(w)}
    clock[w] += 1
    ours.add(step, w)
    ref.add(step, w)
    assert len(ours) == len(ref)
    if len(ref) and n % 7 == 0:
      mode = ('train', 'report')[int(gen.integers(0, 2))]
      got = ours.sample(3, mode)
      want = ref.sample(3, mode)
      assert_same(got, want, f'seed{seed} n{n}')
  got, want = ours.stats(), ref.stats()
  for k in got:
    assert got[k] == want[k], k


def test_pool_full_is_reported_before_any_change():
  rep = HostReplay(length=2, capacity=None, chunksize=2, n_slots=2)
  for t in range(2):
    rep.add({'t': np.int32(t)}, 0)      # fills slot 0, opens slot 1
  rep.add({'t': np.int32(2)}, 0)
  before = len(rep)
  with pytest.raises(_lib.PoolFull):
    rep.add({'t': np.int32(3)}, 0)      # would need a third slot
  assert len(rep) == before
  api.emb_replay_grow(rep.h, 4, None)
  rep.pool = {k: np.concatenate([v, np.zeros_like(v)]) for k, v in rep.pool.items()}
  rep.n_slots = 4
  rep.add({'t': np.int32(3)}, 0)
  assert len(rep) == before + 1


def test_sampletree_shape_like_reference_tests():
  """tests/test_sampletree.py:18-58 re-expressed against the native tree."""
  for branching in (2, 3, 5, 10):
    for inserts in (1, 2, 10, 100):
      tree = selectors.SampleTree(branching)
      for k in range(inserts):
        tree.insert(k, 1)
      depths, _ = tree.shape()
      target = max(1, int(np.ceil(np.log(inserts) / np.log(branching))))
      assert len(depths) == inserts and (depths == target).all()
    tree = selectors.SampleTree(branching)
    assert tree.shape()[1] == 1
    gen = np.random.default_rng(0)
    for k in gen.permutation(100):
      tree.insert(int(k), 1)
    nodes = tree.shape()[1]
    for k in gen.permutation(100):
      tree.remove(int(k))
    assert tree.shape()[1] == 1 and len(tree) == 0
    for k in gen.permutation(100):
      tree.insert(int(k), 1)
    assert tree.shape()[1] == nodes
    total = selectors.SampleTree(branching)
    for k in range(50):
      assert total.total == sum(range(k))
      total.insert(k, k)


@pytest.mark.filterwarnings('ignore:invalid value encountered:RuntimeWarning')   # the degenerate NaN case
@pytest.mark.parametrize('seed', range(8))
def test_random_prioritized_histories_against_oracle(seed):
  """Fuzz of the Prioritized selector under a Replay: random exponent / maxfrac
  / initial (finite, inf) / zero_on_sample, interleaved workers, evictions,
  priority write-backs (also onto evicted steps) -- the drawn windows must be
  the oracle's, which is pinned to the reference by `sel_prioritized` and
  `replay_prioritized`."""
  gen = np.random.default_rng(100 + seed)
  length = int(gen.integers(1, 7))
  chunksize = int(gen.integers(2, 12))
  capacity = int(gen.integers(4, 40))
  workers = int(gen.integers(1, 4))
  kw = dict(exponent=float(gen.choice([1.0, 0.8, 0.5])), maxfrac=float(gen.choice([0.0, 0.5, 1.0])),
            initial=float(gen.choice([1.0, np.inf, 0.3])), zero_on_sample=bool(gen.integers(0, 2)),
            branching=int(gen.choice([2, 3, 16])), seed=seed)
  ours = HostReplay(length, capacity, chunksize, False, selector=selectors.Prioritized(**kw), n_slots=256)
  ref = np_oracle.Replay(length, capacity, chunksize, False, selector=np_oracle.Prioritized(**kw))
  clock = [0] * workers
  kept = []
  for n in range(500):
    w = int(gen.integers(0, workers))
    step = {'t': np.int32(clock[w]), 'w': np.int32(w)}
    clock[w] += 1
    ours.add(step, w)
    ref.add(step, w)
    assert len(ours) == len(ref)
    if len(ref) and n % 5 == 0:
      try:
        want = ref.sample(3)
      except ValueError as e:
        # Degenerate settings (e.g. maxfrac 1 with initial inf and zeroing:
        # 0 * inf) turn tree masses into NaN; numpy's choice then refuses the
        # probabilities in the reference.  Same refusal here, then stop.
        assert 'NaN' in str(e), e
        with pytest.raises(ValueError, match='NaN'):
          ours.sample(3)
        return
      got = ours.sample(3)
      assert_same(got, want, f'seed{seed} n{n} {kw}')
      kept.append(want['stepid'])
      if gen.random() < 0.7:
        stepid = kept[int(gen.integers(0, len(kept)))]           # often stale: partly evicted
        prio = gen.random(stepid.shape[:2]) * float(gen.choice([1.0, 10.0, 0.0]))
        ours.update({'stepid': stepid, 'priority': prio})
        ref.update({'stepid': stepid, 'priority': prio})


@pytest.mark.filterwarnings('ignore:invalid value encountered:RuntimeWarning')
def test_prioritized_protocol_with_priorities_for_steps_in_no_item():
  """The selector protocol alone (selectors.py:128-197), with priorities for ANY
  step id -- also ids that belong to no item yet, which the reference's table
  keeps (`prios` is a defaultdict, :139,147) for an item that arrives later:
  the library against the oracle and, in the build container, against the real
  reference class, draw for draw (tools/fuzz_prioritized.py --protocol runs the
  same histories by the thousand)."""
  from tools import fuzz_prioritized
  others = fuzz_prioritized.reference_classes()
  drawn = sum(fuzz_prioritized.protocol(seed, 300, others) for seed in range(60))
  assert drawn > 3000


@pytest.mark.parametrize('seed', range(6))
@pytest.mark.parametrize('break_at', [None, 40, 140])
def test_prioritized_window_streams_then_arbitrary_use(seed, break_at):
  """The selector alone.  Sliding windows over a few streams with oldest-first
  removal (the representation a Replay gets), then -- from `break_at` on --
  removals in random order and items with arbitrary step lists, which force
  the switch to the general representation in mid-history.  Draws and lengths
  must stay the oracle's throughout."""
  gen = np.random.default_rng(500 + seed)
  n = int(gen.integers(2, 6))
  kw = dict(exponent=float(gen.choice([1.0, 0.7])), maxfrac=float(gen.choice([0.0, 0.3])),
            initial=float(gen.choice([1.0, 2.5])), zero_on_sample=bool(gen.integers(0, 2)),
            branching=int(gen.choice([2, 4, 16])), seed=seed)
  ours, ref = selectors.Prioritized(**kw), np_oracle.Prioritized(**kw)
  sid = lambda stream, t: np.frombuffer(
      int(stream + 1).to_bytes(16, 'big') + int(t).to_bytes(4, 'big'), np.uint8)
  clock = [0, 0, 0]
  live, key = [], 0
  for it in range(220):
    free = break_at is not None and it >= break_at
    op = gen.random()
    if op < 0.5 or len(live) < 3:
      stream = int(gen.integers(0, 3))
      if free and gen.random() < 0.3:       # arbitrary item: steps from anywhere, repeats allowed
        steps = [sid(int(gen.integers(0, 3)), int(gen.integers(0, max(clock) + 2))) for _ in range(n)]
      else:
        t0 = clock[stream]
        clock[stream] += 1
        steps = [sid(stream, t0 + i) for i in range(n)]
      ours[key] = np.stack(steps)
      ref[key] = [s.tobytes() for s in steps]
      live.append((key, stream))
      key += 1
    elif op < 0.7:
      if free:
        victim = live.pop(int(gen.integers(0, len(live))))[0]
      else:                                   # oldest item of some stream
        stream = live[int(gen.integers(0, len(live)))][1]
        index = next(i for i, (_, s) in enumerate(live) if s == stream)
        victim = live.pop(index)[0]
      del ours[victim]
      del ref[victim]
    elif op < 0.85:
      stream = int(gen.integers(0, 3))
      t0 = int(gen.integers(0, clock[stream] + n + 1))
      ids = np.stack([sid(stream, t0 + i) for i in range(int(gen.integers(1, 2 * n)))])
      prios = gen.random(len(ids)) * float(gen.choice([1.0, 5.0, 0.0]))
      ours.prioritize(ids, prios)
      ref.prioritize([x.tobytes() for x in ids], list(prios))
    assert len(ours) == len(ref)
    if len(ref) and it % 3 == 0:
      assert ours() == ref(), (seed, break_at, it)


def test_prioritized_long_windows_against_oracle():
  """Windows long enough (L = 20) for the eight-items-at-a-time aggregation and
  for windows that span chunk boundaries, with zero-on-sample and priority
  write-backs."""
  kw = dict(exponent=0.8, maxfrac=0.5, initial=np.inf, zero_on_sample=True, seed=3)
  ours = HostReplay(20, 150, 16, False, selector=selectors.Prioritized(**kw), n_slots=256)
  ref = np_oracle.Replay(20, 150, 16, False, selector=np_oracle.Prioritized(**kw))
  gen = np.random.default_rng(9)
  for t in range(140):
    for w in range(3):
      step = {'t': np.int32(t), 'w': np.int32(w)}
      ours.add(step, w)
      ref.add(step, w)
    if t >= 25 and t % 6 == 0:
      got, want = ours.sample(4), ref.sample(4)
      assert_same(got, want, f't{t}')
      prio = gen.random(want['stepid'].shape[:2]) * 3
      ours.update({'stepid': want['stepid'], 'priority': prio})
      ref.update({'stepid': want['stepid'], 'priority': prio})


@pytest.mark.parametrize('seed', range(6))
def test_sampletree_random_operations_against_oracle(seed):
  """Fuzz of the b-ary sum tree alone: inserts, removals (tail and interior),
  mass updates incl. 0 and inf, draws; root sum, leaf depths and every draw
  must match the oracle's tree (pinned by the golden `sel_sampletree`)."""
  gen = np.random.default_rng(900 + seed)
  branching = int(gen.choice([2, 3, 5, 16]))
  ours, ref = selectors.SampleTree(branching, seed), np_oracle.SampleTree(branching, seed)
  live, key = [], 0
  mass = lambda: float(gen.choice([gen.random(), gen.random() * 50, 0.0, np.inf], p=[0.6, 0.3, 0.07, 0.03]))
  for it in range(400):
    op = gen.random()
    if op < 0.5 or len(live) < 2:
      m = mass()
      ours.insert(key, m)
      ref.insert(key, m)
      live.append(key)
      key += 1
    elif op < 0.75:
      victim = live.pop(int(gen.integers(0, len(live))) if gen.random() < 0.7 else -1)
      ours.remove(victim)
      ref.remove(victim)
    else:
      k, m = live[int(gen.integers(0, len(live)))], mass()
      ours.update(k, m)
      ref.update(k, m)
    assert len(ours) == len(ref)
    root = ref.root.mass
    assert ours.total == root or (np.isnan(root) and np.isnan(ours.total)), (seed, it)
    if live and it % 4 == 0 and not np.isnan(root):
      assert ours.sample() == ref.sample(), (seed, it)
  depths, _ = ours.shape()
  def leaf_depths(node, d=0):
    kids = getattr(node, 'kids', None)
    return [d] if kids is None else [x for kid in kids for x in leaf_depths(kid, d + 1)]
  assert sorted(depths.tolist()) == sorted(leaf_depths(ref.root)) or not live


@pytest.mark.parametrize('seed', range(4))
def test_random_mixture_histories_against_oracle(seed):
  """Replay over Mixture(uniform, priority): member choice, both members'
  bookkeeping under eviction, priority feedback reaching the prioritized one."""
  gen = np.random.default_rng(300 + seed)
  length, chunksize = int(gen.integers(2, 6)), int(gen.integers(3, 10))
  capacity, workers = int(gen.integers(6, 30)), int(gen.integers(1, 4))
  frac = float(gen.choice([0.25, 0.5, 0.75]))
  pkw = dict(exponent=0.8, maxfrac=0.5, initial=1.0, zero_on_sample=bool(seed % 2), seed=seed + 2)
  mine = selectors.Mixture(
      dict(uniform=selectors.Uniform(seed + 1), priority=selectors.Prioritized(**pkw)),
      dict(uniform=frac, priority=1 - frac), seed=seed + 3)
  theirs = np_oracle.Mixture(
      dict(uniform=np_oracle.Uniform(seed + 1), priority=np_oracle.Prioritized(**pkw)),
      dict(uniform=frac, priority=1 - frac), seed=seed + 3)
  ours = HostReplay(length, capacity, chunksize, False, selector=mine, n_slots=256)
  ref = np_oracle.Replay(length, capacity, chunksize, False, selector=theirs)
  clock = [0] * workers
  for n in range(400):
    w = int(gen.integers(0, workers))
    step = {'t': np.int32(clock[w]), 'w': np.int32(w)}
    clock[w] += 1
    ours.add(step, w)
    ref.add(step, w)
    if len(ref) and n % 4 == 0:
      got, want = ours.sample(2), ref.sample(2)
      assert_same(got, want, f'seed{seed} n{n}')
      if gen.random() < 0.6:
        prio = gen.random(want['stepid'].shape[:2]) * 4
        ours.update({'stepid': want['stepid'], 'priority': prio})
        ref.update({'stepid': want['stepid'], 'priority': prio})


@pytest.mark.parametrize('branching', [2, 3, 5, 10])
def test_sampletree_draw_statistics_like_reference_tests(branching):
  """tests/test_sampletree.py:60-173 re-expressed against the native tree: the
  only survivor is always drawn; equal masses are drawn about equally at any
  scale; draw frequencies follow the masses after inserts and after updates;
  zero-mass entries are never drawn unless every entry is zero; infinite
  masses are the only ones drawn."""
  import collections
  tree = selectors.SampleTree(branching)
  for key in (12, 123, 42):
    tree.insert(key, 1.0)
  tree.remove(12)
  tree.remove(42)
  assert {tree.sample() for _ in range(10)} == {123}

  for inserts in (2, 10):
    for mass in (1e-5, 1.0, 1e5):
      tree = selectors.SampleTree(branching, seed=0)
      keys = list(range(inserts))
      for key in keys:
        tree.insert(key, mass)
      for key in keys[::3]:
        tree.remove(key)
      keys = [k for k in keys if k % 3]
      counts = collections.Counter(tree.sample() for _ in range(100 * len(keys)))
      assert set(counts) == set(keys)
      assert all(c / (100 * len(keys)) > 0.5 / len(keys) for c in counts.values())

  masses = {0: 0, 1: 3, 2: 1, 3: 1, 4: 2, 5: 2}
  total = sum(masses.values())
  def check(tree):
    counts = collections.Counter(tree.sample() for _ in range(100 * len(masses)))
    assert 0 not in counts and counts
    for key, count in counts.items():
      assert 0.7 * masses[key] / total < count / (100 * len(masses)) < 1.3 * masses[key] / total
  for scale in (1e-5, 1, 1e5):
    tree = selectors.SampleTree(branching, seed=0)
    for key, mass in masses.items():
      tree.insert(key, scale * mass)
    check(tree)
  tree = selectors.SampleTree(branching, seed=0)
  for key in masses:
    tree.insert(key, 100)
  for key, mass in masses.items():
    tree.update(key, mass)
  check(tree)

  tree = selectors.SampleTree(branching, seed=0)
  for index in range(100):
    tree.insert(index, 1.0 if index % 3 == 0 else 0.0)
  assert all(tree.sample() % 3 == 0 for _ in range(1000))
  tree = selectors.SampleTree(branching, seed=0)
  for index in range(100):
    tree.insert(index, 0.0)
  assert all(0 <= tree.sample() < 100 for _ in range(1000))
  tree = selectors.SampleTree(branching, seed=0)
  for index in range(100):
    tree.insert(index, np.inf if index % 3 == 0 else 1.0)
  assert all(tree.sample() % 3 == 0 for _ in range(1000))


def test_complete_all_is_all_or_nothing():
  """Checkpoint support: closing every worker's open chunk needs one free slot
  per open chunk.  With too few it must fail BEFORE rotating anybody (a save
  that dies half-way would leave some workers rotated), and succeed after the
  pool grew."""
  import ctypes as C
  rep = HostReplay(length=2, capacity=None, chunksize=4, n_slots=5)
  for w in range(3):
    rep.add({'t': np.int32(w)}, w)            # three open chunks, two free slots
  need, free = C.c_int64(), C.c_int64()
  api.emb_replay_open_chunks(rep.h, C.byref(need))
  api.emb_replay_free_slots(rep.h, C.byref(free))
  assert (need.value, free.value) == (3, 2)
  def table():
    n = C.c_int64()
    api.emb_replay_chunks(rep.h, 0, None, None, None, None, None, C.byref(n))
    uid, succ = np.zeros(n.value, np.uint64), np.zeros(n.value, np.uint64)
    fill, slot, tms = (np.zeros(n.value, np.int64) for _ in range(3))
    api.emb_replay_chunks(rep.h, n.value, _lib.ptr(uid), _lib.ptr(succ), _lib.ptr(fill),
                          _lib.ptr(slot), _lib.ptr(tms), C.byref(n))
    return sorted(zip(uid.tolist(), succ.tolist(), fill.tolist()))
  before = table()
  with pytest.raises(_lib.PoolFull):
    api.emb_replay_complete_all(rep.h)
  assert table() == before                     # nothing was rotated
  api.emb_replay_grow(rep.h, 8, None)
  api.emb_replay_complete_all(rep.h)
  after = table()
  assert len(after) == len(before) + 3
  api.emb_replay_open_chunks(rep.h, C.byref(need))
  assert need.value == 0                       # the successors are empty
  api.emb_replay_complete_all(rep.h)           # nothing to close: no slots needed
  assert table() == after


def test_reserved_chunk_serials_are_not_reissued():
  import ctypes as C
  rep = HostReplay(length=2, capacity=None, chunksize=4, n_slots=6)
  api.emb_replay_reserve_uids(rep.h, 41)
  rep.add({'t': np.int32(0)}, 0)
  n = C.c_int64()
  uid = np.zeros(4, np.uint64)
  api.emb_replay_chunks(rep.h, 4, _lib.ptr(uid), None, None, None, None, C.byref(n))
  assert n.value == 1 and uid[0] == 41
  api.emb_replay_reserve_uids(rep.h, 7)        # never moves backwards
  rep.add({'t': np.int32(0)}, 1)
  api.emb_replay_chunks(rep.h, 4, _lib.ptr(uid), None, None, None, None, C.byref(n))
  assert sorted(uid[:2].tolist()) == [41, 42]


def test_prioritized_with_a_tiny_step_id_backlog():
  """The Prioritized selector keeps its step-id table lazily: adds / erases of
  the sliding-window fast path are logged and applied when `prioritize` (or a
  general insert) needs the table, and after EMB_WHERE_BACKLOG unasked entries
  the log is dropped and the table rebuilt from the streams on demand.  With a
  backlog of 7 (read once per process: child process) the rebuild path runs all
  the time; the oracle comparisons must not notice."""
  import os
  import pathlib
  import subprocess
  import sys
  root = pathlib.Path(__file__).resolve().parent.parent
  env = dict(os.environ, EMB_WHERE_BACKLOG='7')
  res = subprocess.run(
      [sys.executable, '-m', 'pytest', 'tests/test_host_index.py', '-q', '-x', '-m', 'not gpu',
       '-k', 'prioritized and not tiny_step_id_backlog or golden or mixture'],
      cwd=root, env=env, capture_output=True, text=True, timeout=900)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
  assert ' passed' in res.stdout



def test_add_index_reports_openings_in_recycled_slots_only():
  """include/embodied_hip.h: `new_chunks_out` counts the chunks a call opened in
  slots an evicted chunk had held (what makes a caller that batches payload
  writes send the waiting rows first); a slot that was never handed out is not
  reported.  One worker, chunksize 4, 6 slots, capacity 5 items of length 2: a
  chunk's successor is opened by the call that fills it (replay.py:100-104); the
  first five successors take fresh slots 1..5, from the sixth on (call 23, back
  in slot 0) every opening recycles."""
  rep = HostReplay(length=2, capacity=5, chunksize=4, n_slots=6)
  workers, rows, sid = np.zeros(1, np.int64), np.zeros(1, np.int32), np.zeros((1, 20), np.uint8)
  opened = C.c_int32()
  reported, slots = [], []
  for t in range(48):
    api.emb_replay_add_index(rep.h, 1, _lib.ptr(workers), _lib.ptr(rows), _lib.ptr(sid), C.byref(opened))
    reported.append(opened.value)
    slots.append(int(rows[0]) // 4)
  assert slots == [(t // 4) % 6 for t in range(48)]
  assert reported == [1 if t >= 23 and t % 4 == 3 else 0 for t in range(48)]
