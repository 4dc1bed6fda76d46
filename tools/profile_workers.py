"""Where a `parallel=True` Driver step goes (shared-memory worker protocol)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb
from embodied_amd.envs import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
device = torch.device('cuda', 0) if torch.cuda.is_available() else None
fns = [(lambda e=e: synthetic.HostSyntheticEnv(e)) for e in range(n)]
driver = emb.Driver(fns, parallel=True, device=device)
act = np.zeros(n, np.int32)


def policy(carry, obs, **kw):
  return carry, {'action': act}, {}


driver.reset()
driver(policy, steps=n * 50)
host = {'action': act, 'reset': np.zeros(n, bool)}
T = {'release': 0.0, 'wait': 0.0, 'total': 0.0}
iters = 500
orig = driver._step_workers
for _ in range(iters):
  t0 = time.perf_counter()
  for key, (_, slab) in driver._act_slab.items():
    slab[...] = host[key]
  driver._seq += 1
  seq = driver._seq
  driver._ctrl[0] = seq
  t1 = time.perf_counter()
  for wake in driver._wake:
    wake.release()
  t2 = time.perf_counter()
  while not (driver._done == seq).all():
    pass
  t3 = time.perf_counter()
  T['release'] += t2 - t1
  T['wait'] += t3 - t2
  T['total'] += t3 - t0
for k, v in T.items():
  print(f'{k:8s} {v / iters * 1e6:8.1f} us')
t0 = time.perf_counter()
driver(policy, steps=n * 300)
print(f'driver step {(time.perf_counter() - t0) / 300 * 1e6:8.1f} us  ({os.cpu_count()} cpus)')
e = synthetic.HostSyntheticEnv(0)
t0 = time.perf_counter()
for _ in range(2000):
  e.step({'reset': False, 'action': 0})
print(f'one env.step {(time.perf_counter() - t0) / 2000 * 1e6:8.1f} us')
driver.close()
