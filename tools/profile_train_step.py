"""cProfile of the single-process train step of bench.py (sample through the
Consec stream + GAE), interleaved with driver steps as in the timed loop."""
import cProfile
import os
import pstats
import sys
import time

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '0')
os.environ.setdefault('EMB_PURE_PYTHON', '1')     # the profiler sees Python frames only
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench

args = bench.parse()
args.capacity = 20000
device = torch.device('cuda', 0)
emb, env, replay, driver, policy = bench.build_path(args, 0, device)
B, T = args.batch, args.length
stream = iter(emb.streams.Consec(
    emb.streams.Stateless(replay.sample, B, 'train'),
    length=T, consec=1, prefix=args.context, strict=True, contiguous=True))
value = torch.randn(B, T + args.context, device=device)
driver.reset()
for _ in range(1500):
  driver(policy, steps=args.envs)


def train_step():
  batch = next(stream)
  return emb.scans.gae(batch['reward'], value, batch['is_last'], batch['is_terminal'], hor=200, lam=0.8)


def loop(n):
  for i in range(n):
    for _ in range(5):
      driver(policy, steps=args.envs)
    train_step()


loop(200)
torch.cuda.synchronize()
# wall time of the train step inside the loop, without a profiler
t_train = 0.0
t0 = time.perf_counter()
for i in range(400):
  for _ in range(5):
    driver(policy, steps=args.envs)
  a = time.perf_counter()
  train_step()
  t_train += time.perf_counter() - a
total = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'train step {t_train / 400 * 1e6:.1f} us; driver step {(total - t_train) / 2000 * 1e6:.1f} us')
prof = cProfile.Profile()
prof.enable()
loop(600)
prof.disable()
torch.cuda.synchronize()
stats = pstats.Stats(prof)
stats.sort_stats('tottime').print_stats(34)
