"""Per-call host cost of the pieces of one device Driver step (GPU box)."""
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
for _ in range(400):
  driver(policy, steps=args.envs)
torch.cuda.synchronize()
n = args.envs
reset = torch.zeros(n, dtype=torch.bool, device=device)
acts = {'reset': reset}
action = torch.zeros(n, dtype=torch.int32, device=device)
workers = list(range(n))


def timeit(name, fn, iters=3000):
  for _ in range(100):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(iters):
    fn()
  host = (time.perf_counter() - t0) / iters * 1e6
  torch.cuda.synchronize()
  total = (time.perf_counter() - t0) / iters * 1e6
  print(f'{name:28s} host {host:7.2f} us   incl. drain {total:7.2f} us', flush=True)


obs = env.step(acts)
trans = {**obs, 'action': action}
timeit('env.step', lambda: env.step(acts))
timeit('obs_stack', lambda: emb.ops.obs_stack(obs['image'], layout='channels_first', dtype=torch.bfloat16, scale=1 / 255))
timeit('mask_actions', lambda: mask_actions(action, obs['is_last']))
timeit('replay.add_batch', lambda: replay.add_batch(trans, workers))
timeit('policy()', lambda: policy((), obs))
timeit('driver step', lambda: driver(policy, steps=n))
timeit('torch.empty x1', lambda: torch.empty((n, 84, 84, 4), dtype=torch.uint8, device=device))
timeit('replay.sample(16)', lambda: replay.sample(16), iters=500)
import numpy as np
rew = torch.randn(16, 65, device=device)
fl = torch.zeros(16, 65, dtype=torch.bool, device=device)
timeit('scans.gae(16,65)', lambda: emb.scans.gae(rew, rew, fl, fl), iters=2000)
