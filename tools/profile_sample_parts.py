"""Host cost of the statements of Replay.sample, one by one (GPU box)."""
import os, sys, time
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
import embodied_amd as emb
from embodied_amd import _lib
from embodied_amd._lib import fast

args = bench.parse(); args.capacity = 20000
device = torch.device('cuda', 0)
_, env, replay, driver, policy = bench.build_path(args, 0, device)
driver.reset()
for _ in range(500):
  driver(policy, steps=args.envs)
self = replay
N = 20000
acc = [0.0] * 8
clock = time.perf_counter
for it in range(N + 200):
  if it == 200:
    acc = [0.0] * 8
  t0 = clock()
  self._lock.acquire()
  self._flush()
  t1 = clock()
  stream = self._stream()
  t2 = clock()
  out, ptrs = self._alloc_batch(16, self.length)
  t3 = clock()
  sid = out['stepid']
  first = sid.__dict__.get('_emb_first')
  if first is None or len(first) != 16 * _lib.STEPID_BYTES:
    first = sid._emb_first = (C.c_uint8 * (16 * _lib.STEPID_BYTES))()
  t4 = clock()
  fast.emb_replay_sample(self._h, 16, 0, ptrs, None, first, stream)
  t5 = clock()
  self._reraise()
  self._lock.release()
  res = self._finish(out)
  t6 = clock()
  del out, res, sid
  t7 = clock()
  for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6), (t6, t7))):
    acc[i] += b - a
torch.cuda.synchronize()
names = ['lock + _flush', '_stream()', '_alloc_batch', 'first-ids buffer', 'emb_replay_sample (C)', '_reraise, unlock, _finish', 'drop the batch']
for n, a in zip(names, acc):
  print(f'{n:28s} {a / N * 1e6:6.2f} us')
print(f'{"sum (incl. 7 clock reads)":28s} {sum(acc) / N * 1e6:6.2f} us')
