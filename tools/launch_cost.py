import time, torch, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from embodied_amd import _lib
from embodied_amd._lib import fast
dev = torch.device('cuda', 0)
act = torch.ones(64, dtype=torch.int32, device=dev); out = torch.empty_like(act)
last = torch.zeros(64, dtype=torch.bool, device=dev)
s = _lib.raw_stream(dev)
a, o, l = act.data_ptr(), out.data_ptr(), last.data_ptr()
for _ in range(1000): fast.emb_mask_actions(a, o, 64, 1, _lib.I32, l, s)
torch.cuda.synchronize()
for rep in range(3):
  t0 = time.perf_counter()
  for _ in range(20000): fast.emb_mask_actions(a, o, 64, 1, _lib.I32, l, s)
  dt = (time.perf_counter() - t0) / 20000
  torch.cuda.synchronize()
  print(f'mask launch via shim: {dt * 1e6:.2f} us per call (HIP_FORCE_DEV_KERNARG={os.environ.get("HIP_FORCE_DEV_KERNARG")})')
x = torch.zeros(64, device=dev)
t0 = time.perf_counter()
for _ in range(20000): x.add_(1.0)
print(f'torch add_: {(time.perf_counter() - t0) / 20000 * 1e6:.2f} us per call'); torch.cuda.synchronize()

# The movers carry ~3.5 KB of kernel arguments (inline row tables): same cost?
import ctypes as C
import numpy as np
import embodied_amd as emb
rep = emb.Replay(length=2, capacity=64, chunksize=16)
rep.add_batch({'x': torch.zeros((4, 8), device=dev), 'is_first': torch.zeros(4, dtype=torch.bool, device=dev),
               'is_last': torch.zeros(4, dtype=torch.bool, device=dev)}, [0, 1, 2, 3])
rows = np.zeros(1, np.int32)
src = torch.zeros((1, 8), device=dev)
ids = (C.c_int32 * 1)(0)
ptrs = (C.c_void_p * 1)(src.data_ptr())
rp = rows.ctypes.data
for _ in range(1000): fast.emb_replay_scatter_rows(rep._h, rp, 1, 1, ids, ptrs, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20000): fast.emb_replay_scatter_rows(rep._h, rp, 1, 1, ids, ptrs, s)
print(f'one-row scatter (3.5 KB of arguments): {(time.perf_counter() - t0) / 20000 * 1e6:.2f} us per call'); torch.cuda.synchronize()
