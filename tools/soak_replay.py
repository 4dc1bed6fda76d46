"""Soak: many adder threads, sampler threads, periodic save/load on one Replay
(the reference's manual soak: embodied/perf/test_replay.py:78-140).  Every
sampled window is checked for consecutiveness and payload integrity."""
import argparse
import os
import sys
import tempfile
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb

p = argparse.ArgumentParser()
p.add_argument('--seconds', type=float, default=30)
p.add_argument('--adders', type=int, default=32)
p.add_argument('--samplers', type=int, default=8)
args = p.parse_args()

tmp = tempfile.mkdtemp()
replay = emb.Replay(length=8, capacity=4096, directory=tmp, chunksize=64, save_wait=True, numpy=True)
running, errors, counts = [True], [], {'adds': 0, 'samples': 0}


def payload(worker, step):
  return ((np.arange(256) + worker * 7 + step * 13) % 251).astype(np.uint8)


def adder(worker):
  step = 0
  try:
    while running[0]:
      replay.add({'step': np.int32(step), 'worker': np.int32(worker), 'data': payload(worker, step)}, worker)
      step += 1
      counts['adds'] += 1
  except Exception as e:
    errors.append(e)


def sampler():
  try:
    while running[0]:
      batch = replay.sample(4)
      for b in range(4):
        steps, worker = batch['step'][b], int(batch['worker'][b, 0])
        assert (steps - steps[0] == np.arange(8)).all(), steps
        assert (batch['worker'][b] == worker).all()
        for t in (0, 7):
          assert (batch['data'][b, t] == payload(worker, int(steps[t]))).all()
      counts['samples'] += 4
  except Exception as e:
    errors.append(e)


threads = [threading.Thread(target=adder, args=(w,)) for w in range(args.adders)]
threads += [threading.Thread(target=sampler) for _ in range(args.samplers)]
[t.start() for t in threads]
start = time.time()
saves = 0
try:
  while time.time() - start < args.seconds and not errors:
    time.sleep(1.0)
    replay.save()
    replay.load()
    saves += 1
finally:
  running[0] = False
  [t.join() for t in threads]
print('adds', counts['adds'], 'samples', counts['samples'], 'save/load cycles', saves,
      'items', len(replay), 'errors', errors[:2])
sys.exit(1 if errors else 0)
