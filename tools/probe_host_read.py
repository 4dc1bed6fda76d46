"""Can a kernel read the Driver's HIP-registered shared-memory slab directly (zero-copy over
PCIe), and how fast, against hipMemcpyAsync of the same bytes in one and in four pieces?"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb
from embodied_amd import _lib
from embodied_amd._lib import api
from multiprocessing import shared_memory

n, rowbytes = 64, 28224
total = n * rowbytes
block = shared_memory.SharedMemory(create=True, size=total)
host = np.ndarray(total, np.uint8, buffer=block.buf)
host[:] = np.random.default_rng(0).integers(0, 255, total, dtype=np.uint8)
whole = torch.from_numpy(host)
assert torch.cuda.cudart().cudaHostRegister(whole.data_ptr(), total, 0) == 0
dev = torch.empty(total, dtype=torch.uint8, device='cuda')
ids = np.arange(n, dtype=np.int32)
stream = _lib.raw_stream(dev.device)

def timeit(name, fn, iters=200):
  for _ in range(20): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(iters):
    fn(); torch.cuda.synchronize()
  print(f'{name:58s} {(time.perf_counter() - t0) / iters * 1e6:8.1f} us (issue + wait)', flush=True)

def kernel_all():
  api.emb_rows_gather(whole.data_ptr(), rowbytes, _lib.ptr(ids), n, dev.data_ptr(), stream)
def kernel_quarters():
  for g in range(4):
    api.emb_rows_gather(whole.data_ptr() + g * 16 * rowbytes, rowbytes, _lib.ptr(ids[:16]), 16,
                        dev.data_ptr() + g * 16 * rowbytes, stream)
def copy_all():
  dev.copy_(whole, non_blocking=True)
def copy_quarters():
  q = total // 4
  for g in range(4):
    dev[g * q:(g + 1) * q].copy_(whole[g * q:(g + 1) * q], non_blocking=True)

try:
  kernel_all(); torch.cuda.synchronize()
  ok = bool((dev.cpu().numpy() == host).all())
  print('kernel reads the registered slab:', ok)
  timeit('rows_gather kernel, 64 rows from the host slab', kernel_all)
  timeit('rows_gather kernel, 4 x 16 rows', kernel_quarters)
except Exception as e:
  print('kernel read failed:', e)
timeit('hipMemcpyAsync (torch copy_), 1.8 MB', copy_all)
timeit('hipMemcpyAsync (torch copy_), 4 x 450 KB', copy_quarters)
def one_quarter():
  q = total // 4
  dev[:q].copy_(whole[:q], non_blocking=True)
timeit('hipMemcpyAsync (torch copy_), 450 KB', one_quarter)
def kernel_one_quarter():
  api.emb_rows_gather(whole.data_ptr(), rowbytes, _lib.ptr(ids[:16]), 16, dev.data_ptr(), stream)
timeit('rows_gather kernel, 16 rows', kernel_one_quarter)
t0 = time.perf_counter()
for _ in range(200): one_quarter()
print(f'host cost of issuing one 450 KB copy_: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us')
torch.cuda.synchronize()
torch.cuda.cudart().cudaHostUnregister(whole.data_ptr())
del whole, host
block.close(); block.unlink()
