"""Chunk files the PRODUCT wrote (tests/golden/product_chunks, a copy of what
tools/write_product_chunks.py saved on the GPU box) read from the other side:
by the format rules of embodied/core/chunk.py:31-33,64-99 everywhere, and by the
real reference's `Replay.load` in the build container.  CPU only.

`tests/golden/ref_chunks` pins reference -> product; this file pins product ->
reference, so a run may move between the two implementations in either
direction with its replay directory."""
import pathlib

import numpy as np
import pytest

from tools import write_product_chunks as scenario

FIXTURE = pathlib.Path(__file__).parent / 'golden' / 'product_chunks'
ALPHABET = '0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'


def unb62(text):
  value = 0
  for char in text:
    value = value * 62 + ALPHABET.index(char)
  return value


def files():
  names = sorted(p.name for p in FIXTURE.glob('*.npz'))
  assert names, 'tests/golden/product_chunks is empty'
  return names


def expected_windows():
  L = scenario.LENGTH
  return {tuple((w, t + i) for i in range(L))
          for w in range(scenario.WORKERS) for t in range(scenario.STEPS - L + 1)}


def test_file_names_and_arrays_follow_the_chunk_format():
  """`{time}-{uuid}-{succ}-{length}.npz` (chunk.py:31-33): a sortable time
  stamp, two 22-character base-62 ids, the row count; every array has `length`
  rows (chunk.py:68); `stepid` = the 16 big-endian bytes of the chunk's id +
  the row index as 4 big-endian bytes (replay.py:92-95)."""
  per_worker = {}
  for name in files():
    stamp, uid, succ, length = name[:-4].split('-')
    assert len(stamp) == 19 and stamp[8] == 'T' and stamp[15] == 'F', stamp
    assert len(uid) == 22 and len(succ) == 22
    with np.load(FIXTURE / name) as f:
      data = {k: f[k] for k in f.keys()}
    assert set(data) == {'step', 'worker', 'vec', 'image', 'is_first', 'is_last', 'stepid'}
    assert all(len(v) == int(length) for v in data.values()), name
    assert data['step'].dtype == np.int32 and data['vec'].dtype == np.float32
    assert data['image'].dtype == np.uint8 and data['image'].shape[1:] == (2, 2, 3)
    assert data['is_first'].dtype == np.bool_ and data['stepid'].dtype == np.uint8
    want = np.stack([np.frombuffer(
        unb62(uid).to_bytes(16, 'big') + i.to_bytes(4, 'big'), np.uint8) for i in range(int(length))])
    assert (data['stepid'] == want).all(), name
    workers = set(data['worker'].tolist())
    assert len(workers) == 1                       # a chunk belongs to one worker stream
    for t, row in zip(data['step'], range(int(length))):
      ref = scenario.step_of(int(data['worker'][row]), int(t))
      for key, value in ref.items():
        assert (data[key][row] == value).all(), (name, key, row)
    per_worker.setdefault(workers.pop(), []).append((unb62(uid), unb62(succ), data['step'].tolist()))
  # the successor ids chain each worker's chunks in step order
  for worker, chunks in per_worker.items():
    chunks.sort(key=lambda c: c[2][0])
    assert sum((c[2] for c in chunks), []) == list(range(scenario.STEPS))
    for a, b in zip(chunks, chunks[1:]):
      assert a[1] == b[0], (worker, a, b)


@pytest.mark.reference
def test_reference_loads_the_products_chunk_files(tmp_path):
  """Build container only: the real reference Replay restores from the
  product-written directory exactly the windows that were inserted, goes on
  adding to it, and saves / reloads the mix."""
  from oracle import refload
  if not refload.available():
    pytest.skip('no /root/reference here')
  import shutil
  ref = refload.load()
  work = tmp_path / 'replay'
  shutil.copytree(FIXTURE, work)
  L = scenario.LENGTH
  reader = ref.replay.Replay(L, capacity=None, directory=str(work), chunksize=scenario.CHUNKSIZE,
                             save_wait=True)
  reader.load()
  want = expected_windows()
  assert len(reader) == len(want)
  got = set()
  for itemid in sorted(reader.items):
    chunkid, index = reader.items[itemid]
    seq = reader._getseq(chunkid, index, concat=True)
    got.add(tuple(zip(seq['worker'].tolist(), seq['step'].tolist())))
    for i in range(L):
      ref_step = scenario.step_of(int(seq['worker'][i]), int(seq['step'][i]))
      assert (seq['vec'][i] == ref_step['vec']).all() and (seq['image'][i] == ref_step['image']).all()
  assert got == want
  # sampling through the reference's own path, with its annotations
  batch = reader.sample(8)
  assert batch['image'].shape == (8, L, 2, 2, 3) and batch['stepid'].shape == (8, L, 20)
  # write-back by step id lands on the loaded rows (replay.py:152-165, 241-263)
  reader.update({'stepid': batch['stepid'], 'vec': np.full_like(batch['vec'], -1.0)})
  again = reader.sample(64)
  hit = {bytes(s) for s in batch['stepid'].reshape(-1, 20)}
  for sid, vec in zip(again['stepid'].reshape(-1, 20), again['vec'].reshape(-1, 3)):
    if bytes(sid) in hit:
      assert (vec == -1.0).all()
  # the reference continues the run: new steps, new chunk files beside the product's
  for t in range(scenario.STEPS, scenario.STEPS + 6):
    for w in range(scenario.WORKERS):
      reader.add({k: v for k, v in scenario.step_of(w, t).items()}, worker=w)
  reader.save()
  fresh = ref.replay.Replay(L, capacity=None, directory=str(work), chunksize=scenario.CHUNKSIZE)
  fresh.load()
  assert len(fresh) == len(want) + scenario.WORKERS * (6 - L + 1)
