"""Host time of the pieces of the first steps after a fence (torch.cuda.synchronize +
a 200 us pause), against the same pieces in the middle of a burst: what the short
form's first two steps pay (bench.py --steps 20)."""
import os
import sys
import time

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
for _ in range(300):
  driver(policy, steps=args.envs)
torch.cuda.synchronize()
replay.carry_publish(True)
names = ['env.step', 'offer', 'policy', 'add_step']
T = [[0.0] * 4 for _ in range(8)]
acts, workers = driver.acts, driver._workers
reps = 300
pause = float(os.environ.get('PAUSE_US', '200')) * 1e-6
for rep in range(reps):
  torch.cuda.synchronize()
  end = time.perf_counter() + pause
  while time.perf_counter() < end:
    pass
  for step in range(8):
    t0 = time.perf_counter()
    obs = env.step(acts); t1 = time.perf_counter()
    replay.offer(obs, workers); t2 = time.perf_counter()
    carry, a, outs = policy((), obs); t3 = time.perf_counter()
    is_last = obs['is_last']
    replay.add_step(obs, a, outs, workers, is_last, False); t4 = time.perf_counter()
    acts = {**a, 'reset': is_last}
    for j, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
      T[step][j] += d
torch.cuda.synchronize()
print(f'pause {pause * 1e6:.0f} us after the sync; host us per piece, steps 1..8 after it')
print(' ' * 10 + ''.join(f'{n:>10s}' for n in names) + '     total')
for step in range(8):
  row = [x / reps * 1e6 for x in T[step]]
  print(f'step {step + 1:<5d}' + ''.join(f'{x:10.2f}' for x in row) + f'{sum(row):10.2f}')
