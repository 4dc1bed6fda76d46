"""Structural invariants of the Replay boundary, checked against a LEDGER of what
was inserted rather than case by case: the cases the reference's own suite pins
(embodied/tests/test_replay.py: lengths under capacity :46-56, windows of one
worker across chunk boundaries :58-73, interleaved and delayed workers :104-150,
save / load counts and chunk files :152-304, threads :306-357 -- its `dataset(1)`
is today's `sample(1)`) become parameter rows of four properties:

  P1  len(replay) is what FIFO eviction over complete windows leaves;
  P2  every sampled window is `length` consecutive steps of ONE worker's stream,
      with the payload that worker inserted at those steps;
  P3  a checkpoint brings back the same number of items, the same chunk files
      and only windows the ledger knows;
  P4  none of this depends on which thread calls.

Needs a GPU (the pool is HBM)."""
import collections
import pathlib
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GRID = [(1, 1, 1), (2, 1, 2), (5, 1, 10), (1, 2, 2), (5, 3, 15), (2, 7, 20)]     # length, workers, capacity
RESTORE = [(1, 1, 128), (3, 10, 128), (5, 100, 128), (5, 25, 2)]                  # length, capacity, chunksize


@pytest.fixture(scope='module')
def Replay():
  import functools
  import embodied_amd
  return functools.partial(embodied_amd.Replay, numpy=True)


class Ledger:
  """What went into a replay: per worker the list of (step, tag) it inserted."""

  def __init__(self, replay, length):
    self.replay, self.length = replay, length
    self.streams = collections.defaultdict(list)

  def add(self, worker, tag=0, **extra):
    step = len(self.streams[worker])
    self.streams[worker].append((step, tag))
    self.replay.add({'step': step, 'worker': worker, 'tag': tag, **extra}, worker)

  def complete_windows(self):
    return sum(max(0, len(s) - self.length + 1) for s in self.streams.values())

  def check(self, seq):
    """P2 for one sampled sequence (a dict of (length, ...) arrays)."""
    steps, workers = seq['step'], seq['worker']
    assert len(steps) == self.length
    assert (workers == workers[0]).all(), workers
    assert (steps == steps[0] + np.arange(self.length)).all(), steps
    stream = self.streams[int(workers[0])]
    assert 0 <= steps[0] and steps[-1] < len(stream)
    assert [int(t) for t in seq['tag']] == [stream[int(s)][1] for s in steps]

  def draw(self, n=10):
    for _ in range(n):
      seq = {k: v[0] for k, v in self.replay.sample(1).items()}
      self.check(seq)
      yield seq


def lockstep(replay, length, workers, steps):
  ledger = Ledger(replay, length)
  for _ in range(steps):
    for w in range(workers):
      ledger.add(w)
  return ledger


# ---------------------------------------------------------------- P1, P2 --

@pytest.mark.parametrize('length,workers,capacity', GRID)
def test_length_follows_fifo_eviction_at_every_step(Replay, length, workers, capacity):
  replay = Replay(length, capacity)
  ledger = Ledger(replay, length)
  for _ in range(30):
    for w in range(workers):
      ledger.add(w)
    assert len(replay) == min(ledger.complete_windows(), capacity)


@pytest.mark.parametrize('length,workers,capacity,chunksize', [(*row, 128) for row in GRID] + [(7, 2, 27, 4)])
def test_windows_stay_inside_one_stream_across_chunks(Replay, length, workers, capacity, chunksize):
  ledger = lockstep(Replay(length, capacity, chunksize=chunksize), length, workers, 30)
  seen = {(int(s['worker'][0]), int(s['step'][0])) for s in ledger.draw(12)}
  # capacity counts ITEMS: nothing older than the newest `capacity` windows comes back
  oldest_start = 30 - length - (capacity - 1) // workers
  assert all(start >= max(0, oldest_start) for _, start in seen)


@pytest.mark.parametrize('length,capacity', sorted({(l, c) for l, _, c in GRID}))
def test_the_first_window_is_the_whole_history(Replay, length, capacity):
  ledger = lockstep(Replay(length, capacity), length, 1, length)
  assert len(ledger.replay) == 1
  assert all(int(s['step'][0]) == 0 for s in ledger.draw(5))


def test_schema_keys_and_shapes(Replay):
  replay = Replay(length=5, capacity=10)
  for _ in range(30):
    replay.add({'image': np.zeros((64, 64, 3)), 'action': np.zeros(12)})
  batch = replay.sample(1)
  assert {k: v.shape for k, v in batch.items()} == {
      'stepid': (1, 5, 20), 'image': (1, 5, 64, 64, 3), 'action': (1, 5, 12)}
  fixed = Replay(length=2, capacity=10)
  fixed.add({'a': np.float32(1), 'b': np.zeros(3, np.int32)})
  with pytest.raises((KeyError, ValueError)):
    fixed.add({'a': np.float32(1)})                                   # the first step fixed the key set ...
  with pytest.raises(ValueError):
    fixed.add({'a': np.float32(1), 'b': np.zeros(4, np.int32)})       # ... and the shapes (chunk.py:43-47)
  fixed.add({'a': 2.0, 'b': [1, 2, 3], 'log/ignored': 5})             # values are cast, log/* dropped
  seq = {k: v[0] for k, v in fixed.sample(1).items()}
  assert seq['a'].dtype == np.float32 and seq['b'].dtype == np.int32
  assert seq['a'].tolist() == [1.0, 2.0] and seq['b'][1].tolist() == [1, 2, 3]


def test_uniform_selector_reaches_every_item(Replay):
  ledger = lockstep(Replay(capacity=20, length=5, seed=0), 5, 1, 7)
  assert len(ledger.replay) == 3
  counts = collections.Counter(int(s['step'][0]) for s in ledger.draw(120))
  assert sorted(counts) == [0, 1, 2] and min(counts.values()) > 20, counts


def test_interleaved_and_uneven_workers(Replay):
  """Workers that alternate, that insert at random turns and that run dry one
  after another: every window still belongs to one of them."""
  pair = Ledger(Replay(length=2, capacity=20), 2)
  for worker in (0, 1, 0, 1):
    pair.add(worker)
  assert {(int(s['worker'][0]), int(s['step'][0])) for s in pair.draw(12)} == {(0, 0), (1, 0)}
  rng = np.random.default_rng(0)
  uneven = Ledger(Replay(length=4, capacity=30), 4)
  for _ in range(40):
    worker = int(rng.integers(0, 3))
    if len(uneven.streams[worker]) < 10:
      uneven.add(worker, tag=int(rng.integers(0, 100)))
  assert {int(s['worker'][0]) for s in uneven.draw(30)} == {0, 1, 2}
  for length, workers, capacity in GRID:
    drained = Ledger(Replay(length, capacity), length)
    alive = list(range(workers))
    while alive:                        # streams of 10 steps end one after another
      worker = alive[int(rng.integers(0, len(alive)))]
      drained.add(worker)
      if len(drained.streams[worker]) == 10:
        alive.remove(worker)
    assert 0 < len(drained.replay) <= capacity
    list(drained.draw(6))


def test_sample_waits_for_the_first_complete_window(Replay):
  replay = Replay(length=3, capacity=10)
  ledger = Ledger(replay, 3)
  got = []
  thread = threading.Thread(target=lambda: got.append(replay.sample(2)))
  thread.start()
  for alive_after in (True, True, False):          # two steps are not a window yet (replay.py:123)
    time.sleep(0.08)
    assert thread.is_alive() and not got
    ledger.add(0)
    if not alive_after:
      thread.join(timeout=5)
  assert not thread.is_alive() and (got[0]['step'] == [[0, 1, 2]] * 2).all()


# -------------------------------------------------------------------- P3 --

def _files(directory):
  return sorted(pathlib.Path(directory).glob('*.npz'))


@pytest.mark.parametrize('workers', [1, 2, 5])
@pytest.mark.parametrize('length,capacity,chunksize', RESTORE)
def test_checkpoint_round_trip(Replay, tmp_path, workers, length, capacity, chunksize):
  capacity *= workers
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize, save_wait=True)
  ledger = lockstep(replay, length, workers, 40)
  items = len(replay)
  assert items == min(ledger.complete_windows(), capacity)
  state = replay.save()
  files = _files(tmp_path)
  again = Replay(length, capacity, directory=tmp_path, chunksize=chunksize)
  again.load(state)
  assert _files(tmp_path) == files                               # loading writes nothing
  if workers == 1 or chunksize >= length:
    assert len(again) == items
  else:
    # The loader takes the newest files until their ESTIMATED item counts reach
    # the capacity (replay.py:326-333, an estimate that assumes every file's
    # successors come along) and counts again over what it took: with windows
    # longer than a chunk and several streams the newest chunk of every stream
    # starts fewer windows than estimated -- at most length - 1 per stream.
    assert items - workers * (length - 1) <= len(again) <= items
  items = len(again)
  ledger.replay = again
  list(ledger.draw(min(items, 25)))
  # ... and the restored replay goes on like a new one would (the loaded windows
  # are items like any other: FIFO eviction continues over both)
  more = int(1.5 * min(chunksize, 40))
  for _ in range(more):
    again.add({'step': 0, 'worker': 99, 'tag': 7}, worker=99)
  assert len(again) == min(items + max(0, more - length + 1), capacity)


@pytest.mark.parametrize('length,capacity,chunksize', [(1, 1, 1), (3, 10, 5), (5, 100, 12)])
def test_chunk_files_hold_exactly_the_live_steps(Replay, tmp_path, length, capacity, chunksize):
  """chunk.py:31-33 names a file {time}-{uuid}-{succ}-{length}.npz; chunks that
  hold no step of a live item are gone (replay.py:181-191), the open chunk is
  saved with the steps it has."""
  steps = 30
  assert not _files(tmp_path)
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize, save_wait=True)
  lockstep(replay, length, 1, steps)
  state = replay.save()
  files = _files(tmp_path)
  fills = [int(f.stem.split('-')[3]) for f in files]
  live_steps = min(capacity + length - 1, steps)
  first_live_chunk = (steps - live_steps) // chunksize
  n_chunks = -(-steps // chunksize)
  assert len(files) == n_chunks - first_live_chunk
  assert sum(fills) == steps - first_live_chunk * chunksize and all(1 <= f <= chunksize for f in fills)
  again = Replay(length, capacity, directory=tmp_path, chunksize=chunksize)
  again.load(state)
  assert len(again) == len(replay) == min(steps - length + 1, capacity)


def test_payload_bytes_survive_a_checkpoint(Replay, tmp_path):
  from tests import scenarios
  replay = Replay(6, 200, directory=tmp_path, chunksize=16, save_wait=True, seed=1)
  for t in range(70):
    for w in range(2):
      replay.add(scenarios.synth_step(t, w), w)
  replay.save()
  again = Replay(6, 200, directory=tmp_path, chunksize=16, seed=1)
  again.load()
  assert len(again) == len(replay)
  for _ in range(20):
    seq = {k: v[0] for k, v in again.sample(1).items()}
    for j in range(6):
      want = scenarios.synth_step(int(seq['step'][j]), int(seq['worker'][j]))
      assert all(np.array_equal(seq[k][j], want[k]) for k in ('image', 'vec', 'reward'))


@pytest.mark.parametrize('length,capacity,chunksize', RESTORE)
def test_loading_into_a_replay_that_kept_running(Replay, tmp_path, length, capacity, chunksize):
  """replay.py:311-360 appends the checkpoint's items to whatever is there; with
  the capacity smaller than the checkpoint only loaded payload is left."""
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize, save_wait=True)
  ledger = Ledger(replay, length)
  for _ in range(30):
    ledger.add(0, tag=13)
  saved_items = len(replay)
  state = replay.save()
  for _ in range(30):
    ledger.add(0, tag=42)
  replay.load(state)
  assert 0 < len(replay) <= capacity
  for _ in range(len(replay)):
    tags = set(replay.sample(1)['tag'][0].tolist())
    assert tags <= {13, 42}
    if saved_items == capacity and chunksize >= length:
      assert tags == {13}                # the loaded items alone fill it: everything older is evicted


# -------------------------------------------------------------------- P4 --

def test_adders_samplers_and_checkpoints_on_their_own_threads(Replay, tmp_path):
  length, capacity = 5, 128
  replay = Replay(length, capacity, directory=tmp_path, chunksize=32, save_wait=True)
  stop, errors = threading.Event(), []

  def adder(worker):
    step = 0
    while not stop.is_set():
      replay.add({'step': step, 'worker': worker}, worker=worker)
      step += 1
      time.sleep(0.001)

  def sampler():
    try:
      while not stop.is_set():
        seq = replay.sample(1)
        assert (seq['step'][0] == seq['step'][0, 0] + np.arange(length)).all()
        assert (seq['worker'][0] == seq['worker'][0, 0]).all()
        time.sleep(0.001)
    except Exception as e:      # reported by the main thread
      errors.append(e)

  threads = [threading.Thread(target=adder, args=(w,)) for w in range(8)]
  threads += [threading.Thread(target=sampler) for _ in range(4)]
  try:
    for t in threads:
      t.start()
    for _ in range(4):
      time.sleep(0.1)
      stats = replay.stats()
      assert stats['inserts'] > 0 and stats['samples'] > 0
      state = replay.save()
      time.sleep(0.1)
      replay.load(state)
  finally:
    stop.set()
    for t in threads:
      t.join()
  assert not errors, errors
  assert len(replay) == capacity
