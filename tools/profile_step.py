"""Where one device Driver step spends host time (statement-level timers)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
import embodied_amd as emb
from embodied_amd.core.driver import mask_actions

args = bench.parse()
args.capacity = 20000
device = torch.device('cuda', 0)
_, env, replay, driver, policy = bench.build_path(args, 0, device)
driver.reset()
for _ in range(300):
  driver(policy, steps=args.envs)
torch.cuda.synchronize()
T = {}
def lap(name, t0):
  t1 = time.perf_counter()
  T[name] = T.get(name, 0.0) + (t1 - t0)
  return t1

iters = 3000
acts = driver.acts
workers = driver._workers
for _ in range(iters):
  t = time.perf_counter()
  obs = env.step(acts); t = lap('env.step', t)
  carry, a, outs = policy((), obs); t = lap('policy', t)
  is_last = obs['is_last']
  trans = {**obs, **a, **outs}; t = lap('dicts', t)
  # the Driver's only consumer is the replay: mask + insert are one launch
  a = replay.add_batch(trans, workers, mask=(tuple(a), is_last)); t = lap('add_batch(mask)', t)
  acts = {**a, 'reset': is_last}
torch.cuda.synchronize()
total = sum(T.values())
for k, v in T.items():
  print(f'{k:16s} {v / iters * 1e6:7.2f} us')
print(f'{"total":16s} {total / iters * 1e6:7.2f} us')
t0 = time.perf_counter()
for _ in range(iters):
  mask_actions(acts['action'], is_last)
print(f'{"separate mask":16s} {(time.perf_counter() - t0) / iters * 1e6:7.2f} us (not on the fused path)')

# inside add_batch: the C call alone
import ctypes as C
import numpy as np
from embodied_amd import _lib
from embodied_amd._lib import api
w = np.ascontiguousarray(workers, np.int64)
ptrs = (C.c_void_p * len(replay._keys))()
keep = []
for name, value in trans.items():
  i = replay._keyid[name]
  ptrs[i] = value.data_ptr()
stream = replay._stream()
t0 = time.perf_counter()
for _ in range(iters):
  api.emb_replay_add(replay._handle, len(w), _lib.ptr(w), ptrs, stream)
torch.cuda.synchronize()
print(f'emb_replay_add C call   {(time.perf_counter() - t0) / iters * 1e6:7.2f} us')
img = obs['image']
out = torch.empty((64, 4, 84, 84), dtype=torch.bfloat16, device=device)
t0 = time.perf_counter()
for _ in range(iters):
  api.emb_obs_stack(img.data_ptr(), None, 64, 7056, 4, 1, _lib.BF16, 1 / 255, 0.0, out.data_ptr(), stream)
torch.cuda.synchronize()
print(f'emb_obs_stack C call    {(time.perf_counter() - t0) / iters * 1e6:7.2f} us')
t0 = time.perf_counter()
for _ in range(iters):
  _lib.raw_stream(device)
print(f'raw_stream()            {(time.perf_counter() - t0) / iters * 1e6:7.2f} us')
t0 = time.perf_counter()
for _ in range(iters):
  img.data_ptr()
print(f'data_ptr()              {(time.perf_counter() - t0) / iters * 1e6:7.2f} us')
