"""Host cost of the pieces of one Dreamer-style train step (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
import embodied_amd as emb

args = bench.parse(); args.capacity = 20000; args.workload = 'dreamer'
device = torch.device('cuda', 0)
_, env, replay, driver, policy = bench.build_path(args, 0, device)
driver.reset()
for _ in range(500):
  driver(policy, steps=args.envs)
stream = iter(emb.streams.Consec(emb.streams.Stateless(replay.sample, 16, 'train'),
                                 length=64, consec=1, prefix=1, strict=True, contiguous=True))
value = torch.randn(16, 65, device=device)
imag_rew = torch.randn(1024, 16, device=device)
imag_flags = torch.zeros(1024, 16, dtype=torch.bool, device=device)

def timeit(name, fn, iters=1000):
  for _ in range(50): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(iters): fn()
  host = (time.perf_counter() - t0) / iters * 1e6
  torch.cuda.synchronize()
  total = (time.perf_counter() - t0) / iters * 1e6
  print(f'{name:30s} host {host:7.2f} us   host+drain {total:7.2f} us', flush=True)

timeit('driver step', lambda: driver(policy, steps=args.envs))
timeit('replay.sample(16)', lambda: replay.sample(16))
timeit('next(Consec(sample))', lambda: next(stream))
b = next(stream)
timeit('lambda_return replay', lambda: emb.scans.lambda_return(
    b['is_last'], b['is_terminal'], b['reward'], None, value, 1 - 1 / 333, 0.95))
timeit('lambda_return imag', lambda: emb.scans.lambda_return(
    imag_flags, imag_flags, imag_rew, None, imag_rew, 1 - 1 / 333, 0.95))
upd = {'stepid': b['stepid'], 'dyn/deter': b['dyn/deter'], 'dyn/stoch': b['dyn/stoch']}
timeit('replay.update', lambda: replay.update(upd))
import cProfile, pstats
prof = cProfile.Profile(); prof.enable()
for _ in range(1000): replay.update(upd)
prof.disable(); pstats.Stats(prof).sort_stats('tottime').print_stats(8)
