"""Soak of the direct xGMI schedule (emb_direct_*): tests/test_gpu_direct_comm_ranks.py's stress worker with
20 000 operations per world size, ranks sharing this box's GPU.   python tools/soak_direct.py"""
import sys, time
sys.path.insert(0, '.')
import torch.multiprocessing as mp
from tests import test_gpu_direct_comm_ranks as T
if __name__ == '__main__':
  for world in (2, 3, 4):
    m = mp.Manager(); out = m.dict()
    t0 = time.time()
    mp.spawn(T._stress_worker, args=(world, T._free_port(), out, False, 20000), nprocs=world, join=True)
    print('world', world, {r: dict(out[r]) for r in range(world)}, round(time.time() - t0, 1), 's', flush=True)
