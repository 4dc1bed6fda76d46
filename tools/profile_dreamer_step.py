"""cProfile of bench.py's Dreamer train step (sample, two lambda-returns, latent
write-back) interleaved with driver steps."""
import cProfile
import os
import pstats
import sys
import time

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '0')
os.environ.setdefault('EMB_PURE_PYTHON', '1')     # the profiler sees Python frames only
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], '--workload', 'dreamer']
import bench

args = bench.parse()
args.capacity = 50000
device = torch.device('cuda', 0)
emb, env, replay, driver, policy = bench.build_path(args, 0, device)
B, T = args.batch, args.length
stream = iter(emb.streams.Consec(
    emb.streams.Stateless(replay.sample, B, 'train'),
    length=T, consec=1, prefix=args.context, strict=True, contiguous=True))
value = torch.randn(B, T + args.context, device=device)
imag_rew = torch.randn(B * T, 16, device=device)
imag_flags = torch.zeros(B * T, 16, dtype=torch.bool, device=device)
driver.reset()
for _ in range(1200):
  driver(policy, steps=args.envs)

parts = {'sample': 0.0, 'ret1': 0.0, 'ret2': 0.0, 'update': 0.0}


def train_step(timed=False):
  t0 = time.perf_counter()
  batch = next(stream)
  t1 = time.perf_counter()
  emb.scans.lambda_return(
      batch['is_last'], batch['is_terminal'], batch['reward'], None, value, 1 - 1 / 333, 0.95)
  t2 = time.perf_counter()
  emb.scans.lambda_return(imag_flags, imag_flags, imag_rew, None, imag_rew, 1 - 1 / 333, 0.95)
  t3 = time.perf_counter()
  replay.update({'stepid': batch['stepid'], 'dyn/deter': batch['dyn/deter'],
                 'dyn/stoch': batch['dyn/stoch']})
  t4 = time.perf_counter()
  if timed:
    parts['sample'] += t1 - t0
    parts['ret1'] += t2 - t1
    parts['ret2'] += t3 - t2
    parts['update'] += t4 - t3


def loop(n, timed=False):
  for i in range(n):
    driver(policy, steps=args.envs)
    train_step(timed)
    train_step(timed)


loop(200)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop(500, True)
total = time.perf_counter() - t0
torch.cuda.synchronize()
print({k: round(v / 1000 * 1e6, 1) for k, v in parts.items()}, 'us per train step; loop',
      round(total / 500 * 1e6, 1), 'us per vector step')
prof = cProfile.Profile()
prof.enable()
loop(500)
prof.disable()
torch.cuda.synchronize()
pstats.Stats(prof).sort_stats('tottime').print_stats(30)
