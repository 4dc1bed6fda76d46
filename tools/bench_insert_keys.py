"""Insert-shaped scatter of 64 rows, by key subset: the image key alone, the
narrow keys alone, all keys.  Under rocprofv3 --kernel-trace the scatter
kernel's launches come in that order, 300 each."""
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
for t in range(500):
  obs = env.step({'reset': reset})
  reset = obs['is_last']
  rep.add_batch({**obs, 'action': act}, list(range(n)))
torch.cuda.synchronize()
keys = rep._keys
print([(k.name, k.rowbytes) for k in keys])
rows = (np.arange(n, dtype=np.int32) * 1024 + 7).astype(np.int32)
filler = torch.zeros(64 << 20, dtype=torch.uint8, device='cuda')
wide = [i for i, k in enumerate(keys) if k.rowbytes >= 2048]
narrow = [i for i, k in enumerate(keys) if k.rowbytes < 2048]
for subset in (wide, narrow, list(range(len(keys)))):
  srcs = [torch.zeros((n, keys[i].rowbytes), dtype=torch.uint8, device='cuda') for i in subset]
  ids = (C.c_int32 * len(subset))(*subset)
  ptrs = (C.c_void_p * len(subset))(*[s.data_ptr() for s in srcs])
  for i in range(300):
    filler.add_(1)
    api.emb_replay_scatter_rows(rep._handle, _lib.ptr(rows), n, len(subset), ids, ptrs, rep._stream())
  torch.cuda.synchronize()
print('done')
