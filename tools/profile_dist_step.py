"""Host time of one multi-rank train step of bench.py, piece by piece, with ONE
rank (`--force-dist`'s world): what the ranks' code path costs before any link
is involved.   python tools/profile_dist_step.py [--comm direct|c10d] [--iters 3000]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                              # noqa: E402


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--comm', default='direct')
  p.add_argument('--iters', type=int, default=3000)
  mine, rest = p.parse_known_args()
  sys.argv = ['bench.py', *rest]
  args = bench.parse()
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', '29519')
  import torch.distributed as dist
  device = torch.device('cuda', 0)
  torch.cuda.set_device(device)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)
  from embodied_amd import distributed as D
  emb, env, replay, driver, policy = bench.build_path(args, 0, device)
  driver.on_step(lambda *a, **k: None)
  B, T, L = args.batch, args.length, args.length + args.context
  driver.reset()
  for _ in range(-(-(args.capacity + L) // args.envs) + L):
    driver(policy, steps=args.envs)
  grads = torch.zeros(args.grad_numel, dtype=torch.bfloat16, device=device)
  value = torch.randn(B, T + args.context, device=device)
  link = (D.DirectComm(0, 1, device, max_grad_bytes=grads.numel() * 2, max_slice_bytes=64 << 20)
          if mine.comm == 'direct' else D.GroupComm())
  laps = {}

  def lap(name, t0):
    t1 = time.perf_counter_ns()
    laps.setdefault(name, []).append(t1 - t0)
    return time.perf_counter_ns()

  keep, ring, marks = [], [], []
  outs = [tuple(torch.empty(B, T + args.context - 1, device=device) for _ in range(2)) for _ in range(4)]
  for it in range(mine.iters):
    for _ in range(5):
      driver(policy, steps=args.envs)
    t = time.perf_counter_ns()
    pending = replay.online_pending() > 0
    t = lap('online_pending', t)
    flat, batch, layout = D.sample_packed(replay, B, groups=1, reuse=4)
    t = lap('sample_packed', t)
    link.wait()
    t = lap('link.wait', t)
    if it % 3 == 0:
      if not ring:
        ring.extend(torch.empty_like(flat) for _ in range(4))
      received = ring[it // 3 & 3]
      link.exchange(flat, received, grads)
      t = lap('exchange(slices+grads)', t)
      keep[:] = [flat, received]
    else:
      link.exchange(grads=grads)
      t = lap('exchange(grads)', t)
    adv, tar = D.gae_packed(flat, layout, value, hor=200, lam=0.8, out=outs[it & 3])
    t = lap('gae_packed', t)
    if it % 4 == 0:
      if len(marks) == 8:
        mark = marks.pop(0)
        mark.synchronize()
      else:
        mark = torch.cuda.Event()
      mark.record()
      marks.append(mark)
      t = lap('mark', t)
    if it % 64 == 0:
      torch.cuda.synchronize()
  torch.cuda.synchronize()
  print(f'# host microseconds per call, one rank, --comm {mine.comm} (median / mean over {mine.iters} train steps)')
  for name, v in laps.items():
    v = np.asarray(v[len(v) // 10:]) / 1e3
    print(f'{name:28s} {np.median(v):8.2f} {v.mean():8.2f}   n={len(v)}')
  link.close() if hasattr(link, 'close') else None
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
