"""Return scans at the large synthetic size of SURVEY 8d ((65 536, 64)) and at
(1 048 576, 16): run under rocprofv3 --kernel-trace; EMB_SCAN_ROWS=1|2|4 forces
the rows-per-segment variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb

for rows, cols in ((65536, 64), (1048576, 16), (262144, 32)):
  rew = torch.randn(rows, cols, device='cuda')
  val = torch.randn(rows, cols, device='cuda')
  flags = torch.rand(rows, cols, device='cuda') < 0.01
  for _ in range(30):
    emb.scans.gae(rew, val, flags, flags)
    emb.scans.lambda_return(flags, flags, rew, None, val, 0.997, 0.95)
  torch.cuda.synchronize()
print('done')
