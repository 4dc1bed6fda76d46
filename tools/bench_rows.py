"""Sample gather of one synthetic wide key of `rowbytes` bytes per step (plus
the flags): run under rocprofv3 --kernel-trace."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb

rowbytes = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
capacity, n, L = 50_000, 64, 65
rep = emb.Replay(length=L, capacity=capacity, chunksize=1024, seed=0)
wide = torch.zeros(n, rowbytes // 4, dtype=torch.float32, device='cuda')
flag = torch.zeros(n, dtype=torch.bool, device='cuda')
workers = list(range(n))
for t in range(-(-(capacity + L) // n) + L):
  rep.add_batch({'wide': wide, 'is_first': flag, 'is_last': flag, 'is_terminal': flag}, workers)
torch.cuda.synchronize()
for _ in range(300):
  batch = rep.sample(B, 'train')
torch.cuda.synchronize()
print('done', rowbytes, B)
