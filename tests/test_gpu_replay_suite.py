"""What the reference pins in embodied/tests/test_replay.py (structural
invariants: key set/shape, exact lengths under capacity, consecutive
single-worker windows across chunk boundaries, uniformity, interleaved workers,
save/load round trips and chunk file counts, threaded add/sample/save/load),
re-expressed against embodied_amd.Replay; `dataset(1)` of the stale reference
tests becomes `sample(1)`.  Needs a GPU."""
import collections
import pathlib
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def Replay():
  import functools
  import embodied_amd
  return functools.partial(embodied_amd.Replay, numpy=True)


def one(replay):
  return {k: v[0] for k, v in replay.sample(1).items()}


def test_multiple_keys(Replay):
  replay = Replay(length=5, capacity=10)
  for step in range(30):
    replay.add({'image': np.zeros((64, 64, 3)), 'action': np.zeros(12)})
  seq = one(replay)
  assert set(seq.keys()) == {'stepid', 'image', 'action'}
  assert seq['stepid'].shape == (5, 20)
  assert seq['image'].shape == (5, 64, 64, 3)
  assert seq['action'].shape == (5, 12)


@pytest.mark.parametrize('length,workers,capacity', [
    (1, 1, 1), (2, 1, 2), (5, 1, 10), (1, 2, 2), (5, 3, 15), (2, 7, 20)])
def test_capacity_exact(Replay, length, workers, capacity):
  replay = Replay(length, capacity)
  for step in range(30):
    for worker in range(workers):
      replay.add({'step': step}, worker)
    target = min(workers * max(0, (step + 1) - length + 1), capacity)
    assert len(replay) == target


@pytest.mark.parametrize('length,workers,capacity,chunksize', [
    (1, 1, 1, 128), (2, 1, 2, 128), (5, 1, 10, 128), (1, 2, 2, 128),
    (5, 3, 15, 128), (2, 7, 20, 128), (7, 2, 27, 4)])
def test_sample_sequences(Replay, length, workers, capacity, chunksize):
  replay = Replay(length, capacity, chunksize=chunksize)
  for step in range(30):
    for worker in range(workers):
      replay.add({'step': step, 'worker': worker}, worker)
  for _ in range(10):
    seq = one(replay)
    assert (seq['step'] - seq['step'][0] == np.arange(length)).all()
    assert (seq['worker'] == seq['worker'][0]).all()


@pytest.mark.parametrize('length,capacity', [(1, 1), (2, 2), (5, 10), (1, 2), (5, 15), (2, 20)])
def test_sample_single(Replay, length, capacity):
  replay = Replay(length, capacity)
  for step in range(length):
    replay.add({'step': step})
  for _ in range(10):
    assert (one(replay)['step'] == np.arange(length)).all()


def test_sample_uniform(Replay):
  replay = Replay(capacity=20, length=5, seed=0)
  for step in range(7):
    replay.add({'step': step})
  assert len(replay) == 3
  histogram = collections.defaultdict(int)
  for _ in range(100):
    histogram[int(one(replay)['step'][0])] += 1
  assert len(histogram) == 3, histogram
  assert all(count > 20 for count in histogram.values())


def test_workers_simple(Replay):
  replay = Replay(length=2, capacity=20)
  replay.add({'step': 0}, worker=0)
  replay.add({'step': 1}, worker=1)
  replay.add({'step': 2}, worker=0)
  replay.add({'step': 3}, worker=1)
  for _ in range(10):
    assert tuple(one(replay)['step']) in ((0, 2), (1, 3))


def test_workers_random(Replay, length=4, capacity=30):
  rng = np.random.default_rng(seed=0)
  replay = Replay(length, capacity)
  streams = {i: iter(range(10)) for i in range(3)}
  for _ in range(40):
    worker = int(rng.integers(0, 3, ()))
    try:
      replay.add({'step': next(streams[worker]), 'stream': worker}, worker=worker)
    except StopIteration:
      pass
  histogram = collections.defaultdict(int)
  for _ in range(10):
    seq = one(replay)
    assert (seq['step'] - seq['step'][0] == np.arange(length)).all()
    assert (seq['stream'] == seq['stream'][0]).all()
    histogram[int(seq['stream'][0])] += 1
  assert all(count > 0 for count in histogram.values())


@pytest.mark.parametrize('length,capacity,chunksize', [
    (1, 1, 128), (3, 10, 128), (5, 100, 128), (5, 25, 2)])
def test_restore_exact(Replay, tmp_path, length, capacity, chunksize):
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize, save_wait=True)
  for step in range(30):
    replay.add({'step': step})
  num_items = np.clip(30 - length + 1, 0, capacity)
  assert len(replay) == num_items
  data = replay.save()
  replay = Replay(length, capacity, directory=tmp_path)
  replay.load(data)
  assert len(replay) == num_items
  for _ in range(len(replay)):
    seq = one(replay)
    assert len(seq['step']) == length
    assert (seq['step'] - seq['step'][0] == np.arange(length)).all()


@pytest.mark.parametrize('workers', [1, 2, 5])
@pytest.mark.parametrize('length,capacity', [(1, 1), (3, 10), (5, 100)])
def test_restore_workers(Replay, tmp_path, workers, length, capacity):
  capacity *= workers
  replay = Replay(length, capacity, directory=tmp_path, save_wait=True)
  for step in range(50):
    for worker in range(workers):
      replay.add({'step': step, 'worker': worker}, worker)
  num_items = np.clip((50 - length + 1) * workers, 0, capacity)
  assert len(replay) == num_items
  data = replay.save()
  replay = Replay(length, capacity, directory=tmp_path)
  replay.load(data)
  assert len(replay) == num_items
  for _ in range(len(replay)):
    seq = one(replay)
    assert (seq['step'] - seq['step'][0] == np.arange(length)).all()
    assert (seq['worker'] == seq['worker'][0]).all()


@pytest.mark.parametrize('length,capacity,chunksize', [(1, 1, 1), (3, 10, 5), (5, 100, 12)])
def test_restore_chunks_exact(Replay, tmp_path, length, capacity, chunksize):
  assert len(list(pathlib.Path(tmp_path).glob('*.npz'))) == 0
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize, save_wait=True)
  for step in range(30):
    replay.add({'step': step})
  num_items = np.clip(30 - length + 1, 0, capacity)
  assert len(replay) == num_items
  data = replay.save()
  filenames = list(pathlib.Path(tmp_path).glob('*.npz'))
  lengths = [int(x.stem.split('-')[3]) for x in filenames]
  stored_steps = min(capacity + length - 1, 30)
  total_chunks = int(np.ceil(30 / chunksize))
  pruned_chunks = int(np.floor((30 - stored_steps) / chunksize))
  assert len(filenames) == total_chunks - pruned_chunks
  last_chunk_empty = total_chunks * chunksize - 30
  saved_steps = (total_chunks - pruned_chunks) * chunksize - last_chunk_empty
  assert sum(lengths) == saved_steps
  assert all(1 <= x <= chunksize for x in lengths)
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize)
  replay.load(data)
  assert sorted(pathlib.Path(tmp_path).glob('*.npz')) == sorted(filenames)
  assert len(replay) == num_items
  for _ in range(len(replay)):
    assert len(one(replay)['step']) == length


@pytest.mark.parametrize('length,capacity,chunksize', [
    (1, 1, 128), (3, 10, 128), (5, 100, 128), (5, 25, 2)])
def test_restore_insert(Replay, tmp_path, length, capacity, chunksize):
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize, save_wait=True)
  inserts = int(1.5 * chunksize)
  for step in range(inserts):
    replay.add({'step': step})
  num_items = np.clip(inserts - length + 1, 0, capacity)
  assert len(replay) == num_items
  data = replay.save()
  replay = Replay(length, capacity, directory=tmp_path)
  replay.load(data)
  assert len(replay) == num_items
  for step in range(inserts):
    replay.add({'step': step})
  num_items = np.clip(2 * (inserts - length + 1), 0, capacity)
  assert len(replay) == num_items


def test_payload_survives_save_load(Replay, tmp_path):
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
    seq = one(again)
    for j in range(6):
      want = scenarios.synth_step(int(seq['step'][j]), int(seq['worker'][j]))
      for k in ('image', 'vec', 'reward'):
        assert np.array_equal(seq[k][j], want[k])


def test_threading(Replay, tmp_path, length=5, capacity=128, chunksize=32, adders=8, samplers=4):
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize, save_wait=True)
  running = [True]
  errors = []

  def adder():
    ident = threading.get_ident()
    step = 0
    while running[0]:
      replay.add({'step': step}, worker=ident)
      step += 1
      time.sleep(0.001)

  def sampler():
    try:
      while running[0]:
        seq = one(replay)
        assert (seq['step'] - seq['step'][0] == np.arange(length)).all()
        time.sleep(0.001)
    except Exception as e:
      errors.append(e)

  workers = [threading.Thread(target=adder) for _ in range(adders)]
  workers += [threading.Thread(target=sampler) for _ in range(samplers)]
  try:
    [w.start() for w in workers]
    for _ in range(4):
      time.sleep(0.1)
      stats = replay.stats()
      assert stats['inserts'] > 0
      assert stats['samples'] > 0
      data = replay.save()
      time.sleep(0.1)
      replay.load(data)
  finally:
    running[0] = False
    [w.join() for w in workers]
  assert not errors, errors
  assert len(replay) == capacity


def test_sample_blocks_until_the_buffer_has_an_item(Replay):
  """replay.py:123: sampling from an empty buffer waits (limiters.wait) instead
  of failing; it returns as soon as the first full window exists."""
  replay = Replay(length=3, capacity=10)
  got = []
  thread = threading.Thread(target=lambda: got.append(replay.sample(2)))
  thread.start()
  time.sleep(0.1)
  assert thread.is_alive() and not got
  replay.add({'step': 0})
  replay.add({'step': 1})
  time.sleep(0.05)
  assert thread.is_alive()                 # two steps are not a window yet
  replay.add({'step': 2})
  thread.join(timeout=5)
  assert not thread.is_alive()
  assert (got[0]['step'] == [[0, 1, 2], [0, 1, 2]]).all()


def test_schema_is_fixed_by_the_first_step(Replay):
  replay = Replay(length=2, capacity=10)
  replay.add({'a': np.float32(1), 'b': np.zeros(3, np.int32)})
  with pytest.raises((KeyError, ValueError)):
    replay.add({'a': np.float32(1)})
  with pytest.raises(ValueError):
    replay.add({'a': np.float32(1), 'b': np.zeros(4, np.int32)})
  replay.add({'a': 2.0, 'b': [1, 2, 3], 'log/ignored': 5})   # casts like numpy, drops log/*
  seq = one(replay)
  assert seq['a'].dtype == np.float32 and seq['b'].dtype == np.int32
  assert seq['a'].tolist() == [1.0, 2.0] and seq['b'][1].tolist() == [1, 2, 3]


@pytest.mark.parametrize('length,workers,capacity', [
    (1, 1, 1), (2, 1, 2), (5, 1, 10), (1, 2, 2), (5, 3, 15), (2, 7, 20)])
def test_worker_delay(Replay, length, workers, capacity):
  """Workers that finish at different times (tests/test_replay.py:137-150):
  streams run dry one after another while the others keep inserting."""
  replay = Replay(length, capacity)
  rng = np.random.default_rng(seed=0)
  streams = [iter(range(10)) for _ in range(workers)]
  added = 0
  while streams:
    worker = int(rng.integers(0, len(streams)))
    try:
      replay.add({'step': next(streams[worker])}, worker)
      added += 1
    except StopIteration:
      del streams[worker]
  assert added == 10 * workers
  assert 0 < len(replay) <= capacity


@pytest.mark.parametrize('length,capacity,chunksize', [
    (1, 1, 128), (3, 10, 128), (5, 100, 128), (5, 25, 2)])
def test_restore_noclear(Replay, tmp_path, length, capacity, chunksize):
  """Loading a checkpoint into a replay that kept running
  (tests/test_replay.py:177-193): nothing breaks; where the old items must
  have displaced the new ones, only old payload is sampled."""
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize, save_wait=True)
  for _ in range(30):
    replay.add({'foo': 13})
  num_items = np.clip(30 - length + 1, 0, capacity)
  assert len(replay) == num_items
  data = replay.save()
  for _ in range(30):
    replay.add({'foo': 42})
  replay.load(data)
  assert 0 < len(replay) <= capacity
  for _ in range(len(replay)):
    seq = one(replay)
    assert len(seq['foo']) == length and set(np.unique(seq['foo'])) <= {13, 42}
    if capacity < num_items:
      assert (seq['foo'] == 13).all()


@pytest.mark.parametrize('workers', [1, 2, 5])
@pytest.mark.parametrize('length,capacity,chunksize', [(1, 1, 1), (3, 10, 5), (5, 100, 12)])
def test_restore_chunks_workers(Replay, tmp_path, workers, length, capacity, chunksize):
  """Chunk files per worker: how many are written, how many steps they hold
  after pruning, and that loading brings every item back
  (tests/test_replay.py:252-278)."""
  capacity *= workers
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize, save_wait=True)
  for step in range(50):
    for worker in range(workers):
      replay.add({'step': step}, worker)
  num_items = np.clip((50 - length + 1) * workers, 0, capacity)
  assert len(replay) == num_items
  data = replay.save()
  filenames = list(pathlib.Path(tmp_path).glob('*.npz'))
  lengths = [int(x.stem.split('-')[3]) for x in filenames]
  stored_steps = min(capacity // workers + length - 1, 50)
  total_chunks = int(np.ceil(50 / chunksize))
  pruned_chunks = int(np.floor((50 - stored_steps) / chunksize))
  assert len(filenames) == (total_chunks - pruned_chunks) * workers
  last_chunk_empty = total_chunks * chunksize - 50
  saved_steps = (total_chunks - pruned_chunks) * chunksize - last_chunk_empty
  assert sum(lengths) == saved_steps * workers
  replay = Replay(length, capacity, directory=tmp_path, chunksize=chunksize)
  replay.load(data)
  assert len(replay) == num_items
  for _ in range(len(replay)):
    assert len(one(replay)['step']) == length
