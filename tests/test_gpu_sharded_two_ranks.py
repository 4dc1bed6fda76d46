"""ShardedReplay with a REAL collective: two ranks (gloo, 127.0.0.1) sharing the
test box's one GPU.  Every rank must end up with the batch a single replay over
all envs returns."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _worker(rank, world, port, out):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  import embodied_amd as emb
  from embodied_amd import distributed as D
  from tests import scenarios
  torch.cuda.set_device(0)
  D.init('gloo')
  try:
    n, L = 3, 5
    shard = D.ShardedReplay(L, 40, n, chunksize=8, seed=4)
    single = emb.Replay(L, 40, chunksize=8, seed=4) if rank == 0 else None
    for t in range(37):
      steps = [scenarios.synth_step(t, w) for w in range(n * world)]
      stacked = {k: torch.as_tensor(np.stack([s[k] for s in steps])).cuda() for k in steps[0]}
      shard.add_batch({k: v[rank * n:(rank + 1) * n] for k, v in stacked.items()})
      if single is not None:
        single.add_batch(stacked, list(range(n * world)))
    result = []
    for _ in range(3):
      got = {k: v.cpu().numpy() for k, v in shard.sample(7).items()}
      want = ({k: v.cpu().numpy() for k, v in single.sample(7).items()}
              if single is not None else None)
      result.append((got, want))
    out[rank] = result
  finally:
    torch.distributed.destroy_process_group()


def test_sharded_replay_two_ranks_one_gpu():
  manager = mp.Manager()
  out = manager.dict()
  mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
  for i in range(3):
    got0, want = out[0][i]
    got1, _ = out[1][i]
    assert set(got0) == set(want)
    for k in want:
      assert np.array_equal(got0[k], want[k]), k       # rank 0 == single replay
      assert np.array_equal(got1[k], want[k]), k       # rank 1 has the same batch
