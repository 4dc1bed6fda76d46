import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import embodied_amd as emb
for rows, cols in ((16, 64), (1024, 16), (16, 65), (64, 200)):
  rew = torch.randn(rows, cols, device='cuda'); val = torch.randn(rows, cols, device='cuda')
  flags = torch.rand(rows, cols, device='cuda') < 0.01
  for _ in range(100):
    emb.scans.gae(rew, val, flags, flags)
    emb.scans.lambda_return(flags, flags, rew, None, val, 0.997, 0.95)
  torch.cuda.synchronize()
