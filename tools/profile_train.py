"""Host cost of the pieces of one train step (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
import embodied_amd as emb

args = bench.parse(); args.capacity = 20000
device = torch.device('cuda', 0)
_, env, replay, driver, policy = bench.build_path(args, 0, device)
driver.reset()
for _ in range(500):
  driver(policy, steps=args.envs)
stream = iter(emb.streams.Consec(emb.streams.Stateless(replay.sample, 16, 'train'),
                                 length=64, consec=1, prefix=1, strict=True, contiguous=True))
value = torch.randn(16, 65, device=device)

def timeit(name, fn, iters=2000, burst=8):
  # Bursts of a few calls with the GPU drained in between (not timed): back to
  # back, 11 us gathers fill the queue and the launch call then waits for the
  # GPU -- that is back-pressure, not host cost (a train step comes every five
  # env steps in the loop).
  for _ in range(50): fn()
  host = 0.0
  for _ in range(iters // burst):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(burst): fn()
    host += time.perf_counter() - t0
  host = host / (iters // burst * burst) * 1e6
  torch.cuda.synchronize()
  print(f'{name:30s} host {host:7.2f} us', flush=True)

timeit('replay.sample(16)', lambda: replay.sample(16))
timeit('next(Consec(sample))', lambda: next(stream))
b = next(stream)
timeit('scans.gae', lambda: emb.scans.gae(b['reward'], value, b['is_last'], b['is_terminal']))
# as bench.py's train step does it: batches lent for one draw, GAE into agent-owned pairs
lend = iter(emb.streams.Consec(emb.streams.Stateless(replay.sample, 16, 'train', recycle=1),
                               length=64, consec=1, prefix=1, strict=True, contiguous=True))
pairs = [tuple(torch.empty(16, 64, device=device) for _ in range(2)) for _ in range(2)]
turn = [0]
def bench_train_step():
  batch = next(lend)
  turn[0] += 1
  return emb.scans.gae(batch['reward'], value, batch['is_last'], batch['is_terminal'], hor=200, lam=0.8,
                       out=pairs[turn[0] & 1])
timeit('next(Consec(sample, recycle=1))', lambda: next(lend))
held = replay.sample(16)
def sample_recycled():
  replay.recycle(held)
  return replay.sample(16)
timeit('recycle + replay.sample(16)', sample_recycled)
timeit('replay.sample(16, out=held)', lambda: replay.sample(16, out=held))
timeit('scans.gae(out=)', lambda: emb.scans.gae(b['reward'], value, b['is_last'], b['is_terminal'], hor=200,
                                              lam=0.8, out=pairs[0]))
timeit('train step as in bench.py', bench_train_step)
timeit('_alloc_batch', lambda: replay._alloc_batch(16, 65))
timeit('torch.full consec', lambda: torch.full((16, 65), 0, dtype=torch.int32, device=device))
import cProfile, pstats
prof = cProfile.Profile(); prof.enable()
for _ in range(2000): bench_train_step()
prof.disable(); pstats.Stats(prof).sort_stats('tottime').print_stats(14)
