"""Write-back (`Replay.update`) of PPO-shaped rows: sample B sequences, write
the image key back over them.  Run under `rocprofv3 --kernel-trace` and read the
scatter kernel's duration (tools/summarize_trace.py); EMB_SPAN_MOVER=0 sends it
through the flat mover."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb
from embodied_amd.envs import synthetic

capacity, n, L = 100_000, 64, 65
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
env = synthetic.SyntheticBatchEnv(n)
rep = emb.Replay(length=L, capacity=capacity, chunksize=1024, seed=0)
reset = torch.ones(n, dtype=torch.bool, device='cuda')
act = torch.zeros(n, dtype=torch.int32, device='cuda')
workers = list(range(n))
for t in range(-(-(capacity + L) // n) + L):
  obs = env.step({'reset': reset})
  reset = obs['is_last']
  rep.add_batch({**obs, 'action': act}, workers)
torch.cuda.synchronize()
for _ in range(300):
  batch = rep.sample(B, 'train')
  rep.update({'stepid': batch['stepid'], 'image': batch['image']})
torch.cuda.synchronize()
print('done', B)
