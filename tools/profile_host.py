"""cProfile of the bench step loop (host side) on the GPU box."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench

args = bench.parse()
args.capacity = 20000
device = torch.device('cuda', 0)
emb, env, replay, driver, policy = bench.build_path(args, 0, device)
driver.reset()
for _ in range(400):
  driver(policy, steps=args.envs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
  driver(policy, steps=args.envs)
torch.cuda.synchronize()
print('us/step', (time.perf_counter() - t0) / 2000 * 1e6)
prof = cProfile.Profile()
prof.enable()
for _ in range(2000):
  driver(policy, steps=args.envs)
prof.disable()
torch.cuda.synchronize()
pstats.Stats(prof).sort_stats('tottime').print_stats(22)
