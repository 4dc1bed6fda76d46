"""Why is a 20-step timed region slower per step than a long one?  Times every
one of the first steps after a device synchronize (bench.py's fence)."""
import os
import sys
import time

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '0')
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
import embodied_amd as emb

args = bench.parse()
args.capacity = 20000
device = torch.device('cuda', 0)
_, env, replay, driver, policy = bench.build_path(args, 0, device)
driver.reset()
for _ in range(700):
  driver(policy, steps=args.envs)
stream = iter(emb.streams.Consec(emb.streams.Stateless(replay.sample, 16, 'train'),
                                 length=64, consec=1, prefix=1, strict=True, contiguous=True))
value = torch.randn(16, 65, device=device)
should_train = bench.Ratio(3.0 / 1024)
count = [0]

def one_step():
  driver(policy, steps=args.envs)
  count[0] += args.envs
  for _ in range(should_train(count[0])):
    b = next(stream)
    emb.scans.gae(b['reward'], value, b['is_last'], b['is_terminal'], hor=200, lam=0.8)

for _ in range(300):
  one_step()
for trial in range(3):
  torch.cuda.synchronize()
  stamps = [time.perf_counter()]
  for _ in range(40):
    one_step()
    stamps.append(time.perf_counter())
  t_sync = time.perf_counter()
  torch.cuda.synchronize()
  t_end = time.perf_counter()
  us = [(b - a) * 1e6 for a, b in zip(stamps, stamps[1:])]
  print(f'trial {trial}: first 10 steps us: ' + ' '.join(f'{x:.0f}' for x in us[:10]) +
        f' | mean of 40: {sum(us) / 40:.1f} | final synchronize {1e6 * (t_end - t_sync):.0f} us')
