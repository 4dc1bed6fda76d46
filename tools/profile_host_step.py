"""Where one `parallel=True` device-mode Driver step goes (real simulators on the
host: workers in shared memory, one upload per step): the dependency chain
env step -> upload -> policy -> action back -> next env step, phase by phase."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], '--host-envs', '--parallel-envs']
import bench

args = bench.parse()
args.capacity = 20000
device = torch.device('cuda', 0)
emb, env, replay, driver, policy = bench.build_path(args, 0, device)
driver.reset()
for _ in range(300):
  driver(policy, steps=args.envs)
torch.cuda.synchronize()
T = {}
def lap(name, t0):
  t1 = time.perf_counter()
  T[name] = T.get(name, 0.0) + (t1 - t0)
  return t1

iters = 2000
d = driver
from embodied_amd.core.driver import mask_actions
for _ in range(iters):
  t = time.perf_counter()
  host = d._acts_on_host
  if host is not None:
    d._wait_acts()
    d._acts_on_host = None
  else:
    host = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in d.acts.items()}
  t = lap('actions on the host (wait for the store kernel / the copies of the last step)', t)
  d._wait_uploads(); t = lap('wait for the last upload', t)
  results = d._step_workers(host); t = lap('workers: actions down, wake tree, env steps, flags up (+ slab pieces issued)', t)
  obs = d._stack(results); t = lap('stack: host flags (+ whatever of the slab is not up yet)', t)
  d._sinks[0].offer(obs, d._workers); t = lap('offer', t)
  d.carry, acts, outs = policy(d.carry, obs); t = lap('policy (obs stack + early insert)', t)
  is_last = obs['is_last']
  acts = {k: d._to_device(v) for k, v in acts.items()}
  ended = d._host_flags['is_last']
  if ended.any():
    acts = {k: mask_actions(v, is_last) for k, v in acts.items()}
  d.acts = {**acts, 'reset': is_last if d._rotate() else is_last.clone()}
  t = lap('mask / reset clone', t)
  d._fetch_acts()
  if d._upload_pending == 'unrecorded':        # as Driver._step: the actions' event also covers the uploads
    d._upload_pending = False if d._acts_on_host is not None else d._upload_pending
  t = lap('actions towards the host (issue)', t)
  d._dispatch({**obs, **acts, **outs}); t = lap('replay.add_batch (publish)', t)
torch.cuda.synchronize()
total = sum(T.values())
for k, v in T.items():
  print(f'{k:66s} {v / iters * 1e6:7.1f} us')
print(f'{"total":66s} {total / iters * 1e6:7.1f} us  -> {args.envs / (total / iters) / 1e3:.0f} k env steps/s')
# the pieces of the chain in isolation
img = obs['image']
src, (dev, _) = d._upload_src, d._upload_ring[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
  dev.copy_(src, non_blocking=True)
  torch.cuda.synchronize()
print(f'{"H2D of the slab alone, synchronous (PCIe)":66s} {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us for {src.numel() / 1e6:.2f} MB')
a = d.acts['action']
t0 = time.perf_counter()
for _ in range(200):
  a.cpu()
print(f'{"D2H of the action alone, idle GPU":66s} {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us')
driver.close()
