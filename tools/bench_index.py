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
cfg = _lib.ReplayConfig(L, cap, 1024, (cap + L) // 1024 + 130, 1, 0, 0, 1, 0)
h = C.c_void_p()
api.emb_replay_create(C.byref(cfg), None, 0, C.byref(h))
workers = np.arange(n, dtype=np.int64)
rows = np.zeros(n, np.int32)
sids = np.zeros((n, 20), np.uint8)
for _ in range(cap // n + 3 * L):
  api.emb_replay_add_index(h, n, _lib.ptr(workers), _lib.ptr(rows), _lib.ptr(sids), None)
iters = 20000
t0 = time.perf_counter()
for _ in range(iters):
  api.emb_replay_add_index(h, n, _lib.ptr(workers), _lib.ptr(rows), _lib.ptr(sids), None)
dt = (time.perf_counter() - t0) / iters
print(f'add_index({n} workers): {dt * 1e6:.2f} us per call, {dt / n * 1e9:.0f} ns per step')
out = np.zeros((16, L), np.int32)
t0 = time.perf_counter()
for _ in range(iters):
  api.emb_replay_sample_index(h, 16, 1, _lib.ptr(out), None, None)
dt = (time.perf_counter() - t0) / iters
print(f'sample_index(16): {dt * 1e6:.2f} us per call')
api.emb_replay_destroy(h)
