"""Write a directory of chunk files with the PRODUCT's Replay (needs the GPU) for
the reverse direction of the chunk-format pin: tests/golden/product_chunks is a
copy of what this writes, and `tests/test_oracle_golden.py` hands that
directory to the real reference's `Replay.load` (build container only;
embodied/core/replay.py:311-359, chunk.py:77-99).  `tests/golden/ref_chunks`
(oracle/gen_ref_chunks.py) pins reference -> product; this pins product ->
reference.  `tests/test_gpu_checkpoint_edges.py` re-writes the same scenario on
the GPU and compares it with the committed copy (everything but the time stamp).

    python tools/write_product_chunks.py [out_dir]     # default gpurun_out/product_chunks
"""
import pathlib
import shutil
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

LENGTH, CHUNKSIZE, WORKERS, STEPS = 3, 4, 2, 11


def step_of(worker, t):
  """The step of oracle/gen_ref_chunks.py plus an image key (uint8, 3-d)."""
  return {
      'step': np.int32(t), 'worker': np.int32(worker),
      'vec': (np.arange(3, dtype=np.float32) + 10 * t + worker),
      'image': ((np.arange(2 * 2 * 3).reshape(2, 2, 3) + 7 * t + 3 * worker) % 256).astype(np.uint8),
      'is_first': np.bool_(t == 0), 'is_last': np.bool_(t == STEPS - 1),
  }


def write(out, emb=None):
  """Fill a Replay with the scenario and save it into `out` (emptied first)."""
  if emb is None:
    import embodied_amd as emb
  out = pathlib.Path(out)
  if out.exists():
    shutil.rmtree(out)
  out.mkdir(parents=True)
  rep = emb.Replay(length=LENGTH, capacity=None, directory=out, chunksize=CHUNKSIZE,
                   save_wait=True, seed=0)
  for t in range(STEPS):
    for w in range(WORKERS):
      rep.add(step_of(w, t), w)
  rep.save()
  return sorted(p.name for p in out.glob('*.npz'))


if __name__ == '__main__':
  # (argparse, not sys.argv[1]: `--help` once became a directory of that name in the repo root)
  import argparse
  parser = argparse.ArgumentParser(description=__doc__)
  parser.add_argument('target', nargs='?', default=str(ROOT / 'gpurun_out' / 'product_chunks'),
                      help='directory to (re)write the chunk files into')
  for name in write(parser.parse_args().target):
    print(name)
