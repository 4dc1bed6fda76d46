"""cProfile of the vectorised device step (Driver + Replay.add_batch + policy)."""
import cProfile
import os
import pstats
import sys

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '0')
os.environ.setdefault('EMB_PURE_PYTHON', '1')     # the profiler sees Python frames only
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench

args = bench.parse()
args.capacity = 20000
device = torch.device('cuda', 0)
_, env, replay, driver, policy = bench.build_path(args, 0, device)
driver.reset()
for _ in range(1000):
  driver(policy, steps=args.envs)
torch.cuda.synchronize()
prof = cProfile.Profile()
prof.enable()
for _ in range(5000):
  driver(policy, steps=args.envs)
prof.disable()
torch.cuda.synchronize()
stats = pstats.Stats(prof)
stats.sort_stats('tottime').print_stats(18)
