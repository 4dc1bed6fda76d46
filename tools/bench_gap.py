"""Why is the sample gather ~1.5 us slower inside bench.py than in a tight loop?

Runs the bench path with knobs and prints the gather's average launch time
(dispatch begin/end stamps) for each combination.
"""
import argparse
import itertools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import embodied_amd as emb


def run(between, sync, online, scans, iters=400, capacity=100000):
  sys.argv = [sys.argv[0]]
  args = bench.parse()
  args.capacity = capacity
  device = torch.device('cuda', 0)
  _, env, replay, driver, policy = bench.build_path(args, 0, device)
  driver.reset()
  L = args.length + args.context
  for _ in range(-(-(capacity + L) // args.envs) + L):
    driver(policy, steps=args.envs)
  value = torch.randn(args.batch, L, device=device)
  replay.profile(True)
  replay.profile_read(reset=True)
  for i in range(iters):
    for _ in range(between):
      driver(policy, steps=args.envs)
    if sync:
      torch.cuda.synchronize()
    batch = replay.sample(args.batch, 'train' if online else 'report')
    if scans:
      emb.scans.gae(batch['reward'], value, batch['is_last'], batch['is_terminal'], hor=200, lam=0.8)
  torch.cuda.synchronize()
  launches, ms = replay.profile_read(reset=True)
  return ms / max(launches, 1) * 1e3


def grid():
  p = argparse.ArgumentParser()
  p.add_argument('--iters', type=int, default=400)
  a = p.parse_args()
  print('between sync online scans  gather_us')
  for between, sync, online, scans in itertools.product((0, 1, 5), (0, 1), (1, 0), (0, 1)):
    if between == 0 and online:
      continue   # the online queue runs dry without inserts
    us = run(between, sync, online, scans, a.iters)
    print(f'{between:7d} {sync:4d} {online:6d} {scans:5d}  {us:8.2f}', flush=True)



def micro(kind, iters=400, capacity=100000, batch=None):
  """Tight sample loop with one tiny launch of `kind` between gathers."""
  from embodied_amd.core.driver import mask_actions
  sys.argv = [sys.argv[0]]
  args = bench.parse()
  args.capacity = capacity
  if batch:
    args.batch = batch
  device = torch.device('cuda', 0)
  _, env, replay, driver, policy = bench.build_path(args, 0, device)
  driver.reset()
  L = args.length + args.context
  for _ in range(-(-(capacity + L) // args.envs) + L):
    driver(policy, steps=args.envs)
  value = torch.randn(args.batch, L, device=device)
  tiny = torch.zeros(64, device=device)
  act = torch.ones(64, dtype=torch.int32, device=device)
  last = torch.zeros(64, dtype=torch.bool, device=device)
  rew = torch.zeros(16, 65, device=device)
  flags = torch.zeros(16, 65, dtype=torch.bool, device=device)
  big_a = torch.zeros(64 << 20, dtype=torch.uint8, device=device)
  big_b = torch.zeros(64 << 20, dtype=torch.uint8, device=device)
  rows = np.arange(65, dtype=np.int32)[None]
  timed = os.environ.get('GAP_NO_TIMER') != '1'
  replay.profile(timed)
  replay.profile_read(reset=True)
  for i in range(iters):
    batch = replay.sample(args.batch, 'report')
    if kind == 'torch_fill':
      tiny.fill_(1.0)
    elif kind == 'emb_mask':
      mask_actions(act, last)
    elif kind == 'emb_gae_const':
      emb.scans.gae(rew, value, flags, flags, hor=200, lam=0.8)
    elif kind == 'emb_gae_batch':
      emb.scans.gae(batch['reward'], value, batch['is_last'], batch['is_terminal'], hor=200, lam=0.8)
    elif kind == 'tiny_gather':
      replay.profile(False)
      replay.gather(rows)
      replay.profile(timed)
    elif kind == 'torch_copy_64MB':
      big_b.copy_(big_a)
    elif kind == 'sync':
      torch.cuda.synchronize()
    elif kind == 'emb_scatter':          # one vectorised insert (ours, a scatter launch)
      driver(policy, steps=args.envs)
    elif kind == 'fill_then_big':      # tiny launch, then an untimed big gather
      tiny.fill_(1.0)
      replay.profile(False)
      replay.sample(args.batch, 'report')
      replay.profile(timed)
    elif kind == 'fill_then_sleep':
      tiny.fill_(1.0)
      torch.cuda.synchronize()
      time.sleep(200e-6)
  torch.cuda.synchronize()
  launches, ms = replay.profile_read(reset=True)
  return ms / max(launches, 1) * 1e3


if __name__ == '__main__':
  import numpy as np
  if '--grid' in sys.argv:
    sys.argv.remove('--grid')
    grid()
    sys.exit(0)
  if '--kind' in sys.argv:
    kind = sys.argv[sys.argv.index('--kind') + 1]
    batch = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else None
    print(f'{kind:16s} batch {batch} {micro(kind, batch=batch):8.2f} us', flush=True)
    sys.exit(0)
  for kind in ('none', 'sync', 'torch_fill', 'emb_mask', 'emb_gae_const', 'emb_gae_batch',
               'tiny_gather', 'torch_copy_64MB', 'none'):
    print(f'{kind:16s} {micro(kind):8.2f} us', flush=True)
