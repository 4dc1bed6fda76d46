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
iters = 500
t0 = time.perf_counter()
for _ in range(iters):
  driver._step_workers(host)
print(f'_step_workers (actions down, wake, env steps, done flags up) {(time.perf_counter() - t0) / iters * 1e6:8.1f} us')
t0 = time.perf_counter()
driver(policy, steps=n * 300)
print(f'driver step {(time.perf_counter() - t0) / 300 * 1e6:8.1f} us  ({os.cpu_count()} cpus)')
e = synthetic.HostSyntheticEnv(0)
t0 = time.perf_counter()
for _ in range(2000):
  e.step({'reset': False, 'action': 0})
print(f'one env.step {(time.perf_counter() - t0) / 2000 * 1e6:8.1f} us')
driver.close()
