"""Host index core alone (no GPU): emb_replay_add_index for 64 workers per call
and emb_replay_sample_index for 16 windows, at the BASELINE replay shape."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_amd import _lib
from embodied_amd._lib import api

n, L, cap = 64, 65, 100_000
keep = []


def run(kind, iters=3000):
  cfg = _lib.ReplayConfig(L, cap, 1024, (cap + L) // 1024 + 3 * n + 10, 0, 0, 0, 1, 0)
  h, sel = C.c_void_p(), C.c_void_p()
  if kind == 'prioritized':        # ppo/configs.yaml:42: exponent .8, maxfrac .5, initial inf, zero_on_sample
    api.emb_selector_create_prioritized(0.8, float('inf'), 1, 0.5, 16, 0, C.byref(sel))
  if kind == 'mixture':            # ppo/main.py:197-207 with all three fractions on
    from embodied_amd import selectors
    mix = selectors.Mixture(dict(
        uniform=selectors.Uniform(),
        priority=selectors.Prioritized(exponent=0.8, maxfrac=0.5, initial=float('inf'), zero_on_sample=True),
        recency=selectors.Recency(1.0 / np.arange(1, cap + 1) ** 1.0),
    ), dict(uniform=0.5, priority=0.25, recency=0.25))
    keep.append(mix)
    sel = mix._handle
  api.emb_replay_create(C.byref(cfg), sel if kind != 'uniform' else None, 0, C.byref(h))
  workers = np.arange(n, dtype=np.int64)
  rows = np.zeros(n, np.int32)
  sids = np.zeros((n, 20), np.uint8)
  pw, pr, ps = workers.ctypes.data, rows.ctypes.data, sids.ctypes.data
  for _ in range(cap // n + 3 * L):
    api.emb_replay_add_index(h, n, pw, pr, ps, None)
  t0 = time.perf_counter()
  for _ in range(iters):
    api.emb_replay_add_index(h, n, pw, pr, ps, None)
  add = (time.perf_counter() - t0) / iters
  # The same with the caches in the state the real loop leaves them in: the
  # interpreter and the launches between two inserts push the index out of L2.
  junk = np.zeros(16 << 20, np.uint8)
  cold = 0.0
  for _ in range(300):
    junk += 1
    t0 = time.perf_counter()
    api.emb_replay_add_index(h, n, pw, pr, ps, None)
    cold += time.perf_counter() - t0
  cold /= 300
  out = np.zeros((16, L), np.int32)
  po = out.ctypes.data
  t0 = time.perf_counter()
  for _ in range(iters):
    api.emb_replay_sample_index(h, 16, 1, po, None, None)
  draw = (time.perf_counter() - t0) / iters
  print(f'{kind:12s} add_index({n} workers) {add * 1e6:7.2f} us ({add / n * 1e9:4.0f} ns/step; cold caches {cold * 1e6:6.2f} us)   '
        f'sample_index(16) {draw * 1e6:7.2f} us')
  api.emb_replay_destroy(h)


if __name__ == '__main__':
  run('uniform')
  run('prioritized')
  run('mixture')
