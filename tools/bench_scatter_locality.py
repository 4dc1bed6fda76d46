"""Does the insert kernel care where its 64 rows land?  `emb_replay_scatter_rows`
of 64 PPO-shaped steps into (a) 64 adjacent pool rows, (b) rows one chunk
(1024 rows = 29 MB) apart, as the 64 workers' open chunks are.  Run under
rocprofv3 --kernel-trace and compare the scatter kernel's durations by order of
appearance (first 300 launches adjacent, next 300 scattered)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb
from embodied_amd import _lib
from embodied_amd._lib import api
from embodied_amd.envs import synthetic

n, L, capacity = 64, 65, 100_000
env = synthetic.SyntheticBatchEnv(n)
rep = emb.Replay(length=L, capacity=capacity, chunksize=1024, seed=0)
reset = torch.ones(n, dtype=torch.bool, device='cuda')
act = torch.zeros(n, dtype=torch.int32, device='cuda')
workers = list(range(n))
for t in range(2000):
  obs = env.step({'reset': reset})
  reset = obs['is_last']
  rep.add_batch({**obs, 'action': act}, workers)
torch.cuda.synchronize()
keys = rep._keys
srcs = []
for key in keys:
  srcs.append(torch.zeros((n, key.rowbytes), dtype=torch.uint8, device='cuda'))
ids = (C.c_int32 * len(keys))(*range(len(keys)))
ptrs = (C.c_void_p * len(keys))(*[s.data_ptr() for s in srcs])
adjacent = np.arange(n, dtype=np.int32) + 5 * 1024
spread = (np.arange(n, dtype=np.int32) * 1024 + 7).astype(np.int32)
filler = torch.zeros(64 << 20, dtype=torch.uint8, device='cuda')
for rows in (adjacent, spread):
  for i in range(300):
    filler.add_(1)                     # something else on the GPU in between, as in the loop
    api.emb_replay_scatter_rows(rep._handle, _lib.ptr(rows), n, len(keys), ids, ptrs, rep._stream())
  torch.cuda.synchronize()
print('done')
