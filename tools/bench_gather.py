"""Micro-benchmark of the Replay.sample gather kernel: variants x batch sizes.
Usage (GPU box): python tools/bench_gather.py [--capacity 100000]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb
from embodied_amd.envs import synthetic

p = argparse.ArgumentParser()
p.add_argument('--capacity', type=int, default=100_000)
p.add_argument('--iters', type=int, default=50)
p.add_argument('--batches', default='16,64,256')
p.add_argument('--sleep-us', type=float, default=0, help='host idle time between samples')
p.add_argument('--tight', action='store_true',
               help='also time the C entry point with preallocated outputs (no Python allocation '
                    'between launches: the GPU stays busy, as in tools/gather_lab.hip mode A)')
args = p.parse_args()

n, L = 64, 65
env = synthetic.SyntheticBatchEnv(n)
rep = emb.Replay(length=L, capacity=args.capacity, chunksize=1024, seed=0)
reset = torch.ones(n, dtype=torch.bool, device='cuda')
act = torch.zeros(n, dtype=torch.int32, device='cuda')
workers = list(range(n))
for t in range(-(-(args.capacity + L) // n) + L):
  obs = env.step({'reset': reset})
  reset = obs['is_last']
  rep.add_batch({**obs, 'action': act}, workers)
torch.cuda.synchronize()
S = sum(k.rowbytes for k in rep._keys)
print('items', len(rep), 'S', S, flush=True)

# reference: plain device-to-device copy of the same byte count
for B in map(int, args.batches.split(',')):
  nbytes = B * L * S
  a = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
  b = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
  for _ in range(5):
    b.copy_(a)
  e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
  e0.record()
  for _ in range(args.iters):
    b.copy_(a)
  e1.record()
  torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / args.iters * 1e3
  print(f'torch copy_  B={B:4d} {us:8.2f} us  {2 * nbytes / us / 1e3:8.1f} GB/s (r+w)', flush=True)

rep.profile(True)
for variant in ['ship']:
  for B in map(int, args.batches.split(',')):
    for _ in range(5):
      rep.sample(B)
    rep.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.iters):
      rep.sample(B)
      if args.sleep_us:
        t1 = time.perf_counter()
        while (time.perf_counter() - t1) * 1e6 < args.sleep_us:
          pass
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.iters * 1e6
    launches, ms = rep.profile_read(reset=True)
    us = ms / launches * 1e3
    nbytes = 2 * B * L * S
    print(f'variant {variant:5s} B={B:4d} kernel {us:8.2f} us  {nbytes / us / 1e3:8.1f} GB/s (r+w)  '
          f'frac {nbytes / us / 1e3 / 8000:.3f}  wall/call {wall:8.1f} us', flush=True)

if args.tight:
  import ctypes as C
  from embodied_amd import _lib
  from embodied_amd._lib import fast
  for B in map(int, args.batches.split(',')):
    outs = [rep._new_batch((B, L)) for _ in range(4)]
    first = (C.c_uint8 * (B * _lib.STEPID_BYTES))()
    stream = rep._stream()
    mode = _lib.MODES['train']
    for i in range(10):
      fast.emb_replay_sample(rep._h, B, mode, outs[i % 4][1], None, first, stream)
    torch.cuda.synchronize()
    rep.profile_read(reset=True)
    t0 = time.perf_counter()
    for i in range(args.iters * 4):
      fast.emb_replay_sample(rep._h, B, mode, outs[i % 4][1], None, first, stream)
    host = (time.perf_counter() - t0) / (args.iters * 4) * 1e6
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / (args.iters * 4) * 1e6
    launches, ms = rep.profile_read(reset=True)
    us = ms / launches * 1e3
    nbytes = 2 * B * L * S
    print(f'tight   B={B:4d} kernel {us:8.2f} us  {nbytes / us / 1e3:8.1f} GB/s (r+w)  '
          f'frac {nbytes / us / 1e3 / 8000:.3f}  host/call {host:6.1f} us  wall/call {wall:8.1f} us', flush=True)
