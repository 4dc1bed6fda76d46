"""Every kernel of the path at BASELINE sizes and at a size large enough to be
bandwidth-bound; run under `rocprofv3 --kernel-trace --stats` to get per-kernel
durations (the script itself only issues the calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb
from embodied_amd.core.driver import mask_actions
from embodied_amd.envs import synthetic

dev = torch.device('cuda', 0)
iters = 50
n, L, B, T = 64, 65, 16, 64

# --- Driver-side kernels at N=64 and N=4096 envs
for envs in (64, 4096):
  env = synthetic.SyntheticBatchEnv(envs)
  reset = torch.zeros(envs, dtype=torch.bool, device=dev)
  act = torch.ones(envs, dtype=torch.int32, device=dev)
  for _ in range(iters):
    obs = env.step({'reset': reset})
    emb.ops.obs_stack(obs['image'], layout='channels_first', dtype=torch.bfloat16, scale=1 / 255)
    mask_actions(act, obs['is_last'])
  torch.cuda.synchronize()

# --- Replay: insert (N=64), sample (B=16), Dreamer update (B=16,T=64 latents)
rep = emb.Replay(length=L, capacity=20000, chunksize=1024, seed=0)
env = synthetic.SyntheticBatchEnv(n)
reset = torch.ones(n, dtype=torch.bool, device=dev)
deter = torch.zeros((n, 8192), device=dev)
stoch = torch.zeros((n, 32, 64), device=dev)
act = torch.zeros(n, dtype=torch.int32, device=dev)
workers = list(range(n))
for t in range(20000 // n + 2 * L):
  obs = env.step({'reset': reset})
  reset = obs['is_last']
  rep.add_batch({**obs, 'action': act, 'dyn/deter': deter, 'dyn/stoch': stoch}, workers)
for _ in range(iters):
  batch = rep.sample(B)
  rep.update({'stepid': batch['stepid'], 'dyn/deter': batch['dyn/deter'][:, :T],
              'dyn/stoch': batch['dyn/stoch'][:, :T]})
torch.cuda.synchronize()

# --- scans at BASELINE sizes and at (65536, 64)
for rows, cols in ((16, 64), (1024, 16), (65536, 64)):
  rew = torch.randn(rows, cols, device=dev)
  val = torch.randn(rows, cols, device=dev)
  flags = torch.rand(rows, cols, device=dev) < 0.02
  for _ in range(iters):
    emb.scans.gae(rew, val, flags, flags)
    emb.scans.lambda_return(flags, flags, rew, None, val, 0.997, 0.95)
  torch.cuda.synchronize()
for steps, cols in ((16, 1024), (16, 262144)):
  rew = torch.randn(steps - 1, cols, device=dev)
  cont = torch.ones(steps, cols, device=dev)
  val = torch.randn(steps, cols, device=dev)
  for _ in range(iters):
    emb.scans.director_score(rew, cont, val)
  torch.cuda.synchronize()
print('done')
