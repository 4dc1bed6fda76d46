"""Host cost of the multi-rank train step (run with world 1 on one GPU):
`bench.py --force-dist` semantics, pieces timed separately."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
import embodied_amd as emb
from embodied_amd import distributed as D

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29577')
device = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)
args = bench.parse(); args.capacity = 20000
_, env, replay, driver, policy = bench.build_path(args, 0, device)
driver.reset()
for _ in range(500):
  driver(policy, steps=args.envs)
comm = D.CommThread(device)
value = torch.randn(16, 65, device=device)
grads = torch.zeros(10_000_000, device=device)


def timeit(name, fn, iters=1000):
  for _ in range(30): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(iters): fn()
  host = (time.perf_counter() - t0) / iters * 1e6
  torch.cuda.synchronize()
  print(f'{name:34s} host {host:7.2f} us', flush=True)


timeit('sample_packed', lambda: D.sample_packed(replay, 16))
flat, batch, info = D.sample_packed(replay, 16)
timeit('3 lazy views', lambda: [D.PackedViews(flat, info.layout)[k] for k in ('reward', 'is_last', 'is_terminal')])
timeit('gae', lambda: emb.scans.gae(batch['reward'], value, batch['is_last'], batch['is_terminal']))
timeit('torch.empty(gathered)', lambda: torch.empty(flat.numel(), dtype=torch.uint8, device=device))
out = torch.empty(flat.numel(), dtype=torch.uint8, device=device)
timeit('inline all_gather (pg level)', lambda: D.async_all_gather(out, flat).wait())
timeit('inline all_reduce (pg level)', lambda: D.async_all_reduce(grads).wait())
timeit('submit+result all_gather', lambda: comm.submit(lambda: D.async_all_gather(out, flat)).result().wait())
timeit('submit only (x2) + late wait', lambda: None)
pending = []
def step():
  for f in pending: f.result().wait()
  pending.clear()
  pending.append(comm.submit(lambda: D.async_all_gather(out, flat)))
  pending.append(comm.submit(lambda: D.async_all_reduce(grads)))
  for _ in range(5): driver(policy, steps=args.envs)
timeit('5 driver steps + 2 collectives', step, 300)
timeit('5 driver steps alone', lambda: [driver(policy, steps=args.envs) for _ in range(5)], 300)
ev = []
def marks():
  ev.append(torch.cuda.Event()); ev[-1].record()
  if len(ev) > 8: ev.pop(0).synchronize()
timeit('event mark + late sync', marks)
comm.close()
dist.destroy_process_group()
