"""Where one device Driver step spends host time (statement-level timers), for
the Driver's own sequence: env.step -> Replay.offer -> policy (ops.obs_stack,
which takes up the offer: early insert) -> add_batch (publish).
EMB_EARLY_INSERT=0 profiles the plain sequence (no offer)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
import embodied_amd as emb
from embodied_amd.core.driver import mask_actions

early = os.environ.get('EMB_EARLY_INSERT', '1') != '0'
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
unmasked = bool(getattr(env, 'takes_unmasked_actions', False)) and os.environ.get('EMB_CARRY_PUBLISH', '1') != '0'
if unmasked:
  replay.carry_publish(True)
ring = [{} for _ in range(4)]
for it in range(iters):
  if it % 16 == 0:
    # (bursts of 16 steps with the GPU drained in between: since round 4 the GPU
    # sets the pace of a long loop, and a launch call that waits for a full queue
    # would be counted as host time)
    torch.cuda.synchronize()
  t = time.perf_counter()
  obs = env.step(acts); t = lap('env.step', t)
  if early:
    replay.offer(obs, workers); t = lap('offer', t)
  carry, a, outs = policy((), obs); t = lap('policy (obs stack [+ early insert])', t)
  is_last = obs['is_last']
  if unmasked:      # the Driver's sequence for an env that takes unmasked actions
    replay.add_step(obs, a, outs, workers, is_last, False); t = lap('add_step (publish, carried)', t)
    acts = {**a, 'reset': is_last}
  else:
    a = replay.add_step(obs, a, outs, workers, is_last, ring[it & 3]); t = lap('add_step (publish + mask)', t)
    acts = {**a, 'reset': is_last}
  t = lap('next acts dict', t)
torch.cuda.synchronize()
trans = {**obs, **a, **outs}
total = sum(T.values())
for k, v in T.items():
  print(f'{k:40s} {v / iters * 1e6:7.2f} us')
print(f'{"total":40s} {total / iters * 1e6:7.2f} us   (early inserts: {replay.early_inserts}, '
      f'carried: {replay.profile_report("carried")[:2]})')

# The whole Driver step, and the pieces in isolation
spent = 0.0
for burst in range(iters // 16):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(16):
    driver(policy, steps=args.envs)
  spent += time.perf_counter() - t0
torch.cuda.synchronize()
print(f'{"driver(policy, steps=N), bursts of 16":40s} {spent / (iters // 16 * 16) * 1e6:7.2f} us')

import ctypes as C
import numpy as np
from embodied_amd import _lib
from embodied_amd._lib import api, fast
w = np.ascontiguousarray(workers, np.int64)
ptrs = (C.c_void_p * len(replay._keys))()
for name, value in trans.items():
  ptrs[replay._keyid[name]] = value.data_ptr()
stream = replay._stream()
def timed(name, fn, n=iters):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  dt = time.perf_counter() - t0
  torch.cuda.synchronize()
  print(f'{name:40s} {dt / n * 1e6:7.2f} us')
timed('emb_replay_add C call', lambda: fast.emb_replay_add(replay._h, len(w), _lib.ptr(w), ptrs, stream))
img = obs['image']
out = torch.empty((64, 4, 84, 84), dtype=torch.bfloat16, device=device)
timed('emb_obs_stack C call', lambda: fast.emb_obs_stack(
    img.data_ptr(), None, 64, 7056, 4, 1, _lib.BF16, 1 / 255, 0.0, out.data_ptr(), stream))
spec = _lib.ObsSpec(7056, 4, 1, _lib.BF16, 1 / 255, 0.0)
token = C.c_uint64()
obs_ptrs = (C.c_void_p * len(replay._keys))()
for name, value in obs.items():
  obs_ptrs[replay._keyid[name]] = value.data_ptr()
ids = (C.c_int32 * 1)(replay._keyid['action'])
codes = (C.c_int32 * 1)(_lib.I32)
outs_ = (C.c_void_p * 1)(a['action'].data_ptr())
def pair():
  fast.emb_replay_obs_stack_insert(
      replay._h, 64, _lib.ptr(w), replay._keyid['image'], img.data_ptr(), C.addressof(spec),
      out.data_ptr(), obs_ptrs, stream, token)
  fast.emb_replay_publish(replay._h, 64, _lib.ptr(w), ptrs, 1, ids, codes, outs_, is_last.data_ptr(),
                          token.value, stream)
timed('obs_stack_insert + publish C calls', pair)
timed('  obs_stack_insert alone (then plain add)', lambda: (fast.emb_replay_obs_stack_insert(
      replay._h, 64, _lib.ptr(w), replay._keyid['image'], img.data_ptr(), C.addressof(spec),
      out.data_ptr(), obs_ptrs, stream, token)))
timed('separate mask kernel', lambda: mask_actions(acts['action'], is_last))
timed('raw_stream()', lambda: _lib.raw_stream(device))
timed('data_ptr()', lambda: img.data_ptr())
timed('tuple(obs)', lambda: tuple(obs))
timed('replay.offer', lambda: replay.offer(obs, workers))
timed('_collect (7 keys)', lambda: replay._collect(trans, replay._add_plan[2], ptrs))
