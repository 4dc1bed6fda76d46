"""Write a directory of chunk files with the REAL reference Replay (build
container only: needs /root/reference) and record what the reference's own
`load()` makes of it.  Output: tests/golden/ref_chunks/*.npz (the chunk files,
named `{time}-{uuid}-{succ}-{length}.npz` with random 128-bit UUIDs, exactly as
embodied/core/chunk.py:31-33,64-75 writes them) and
tests/golden/ref_chunks_expected.npz (item count and every item's window).

TEST INFRASTRUCTURE ONLY.  The fixture pins `Replay.load` of the product against
reference-written files (uids that are not `replica << 64 | serial`).

    python oracle/gen_ref_chunks.py
"""
import pathlib
import shutil
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import refload  # noqa: E402

LENGTH, CHUNKSIZE, WORKERS, STEPS = 3, 4, 2, 11


def step_of(worker, t):
  return {
      'step': np.int32(t), 'worker': np.int32(worker),
      'vec': (np.arange(3, dtype=np.float32) + 10 * t + worker),
      'is_first': np.bool_(t == 0), 'is_last': np.bool_(t == STEPS - 1),
  }


def main():
  ref = refload.load()
  import elements
  elements.UUID.reset(debug=False)          # random 128-bit ids, as in a real run
  out = ROOT / 'tests' / 'golden' / 'ref_chunks'
  if out.exists():
    shutil.rmtree(out)
  out.mkdir(parents=True)
  writer = ref.replay.Replay(LENGTH, capacity=None, directory=str(out), chunksize=CHUNKSIZE,
                             save_wait=True)
  for t in range(STEPS):
    for w in range(WORKERS):
      writer.add(step_of(w, t), worker=w)
  writer.save()
  names = sorted(p.name for p in out.glob('*.npz'))
  # What the reference itself restores from these files.
  reader = ref.replay.Replay(LENGTH, capacity=None, directory=str(out), chunksize=CHUNKSIZE)
  reader.load()
  windows = []
  for itemid in sorted(reader.items):
    chunkid, index = reader.items[itemid]
    seq = reader._getseq(chunkid, index, concat=True)
    windows.append(np.stack([seq['worker'], seq['step']], -1))
  np.savez_compressed(
      ROOT / 'tests' / 'golden' / 'ref_chunks_expected.npz',
      names=np.array(names), items=np.int64(len(reader)),
      windows=np.stack(windows).astype(np.int32), length=np.int64(LENGTH),
      chunksize=np.int64(CHUNKSIZE))
  print(f'{len(names)} chunk files, {len(reader)} items restored by the reference')


if __name__ == '__main__':
  main()
