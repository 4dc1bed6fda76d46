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


def _crafter_worker(rank, world, port, out):
  """configs[3]: 256 envs in 8 blocks of 32 (one per rank), 64x64x3 frames,
  L = 65; the packed batches are merged by a REAL all-reduce between 8 processes
  (gloo on 127.0.0.1; they share the test box's one GPU)."""
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  import hashlib
  import embodied_amd as emb
  from embodied_amd import distributed as D
  torch.cuda.set_device(0)
  D.init('gloo')
  try:
    per, L, B, cap = 32, 65, 16, 6000
    kw = dict(chunksize=256, seed=7)
    shard = D.ShardedReplay(L, cap, per, **kw)
    single = emb.Replay(L, cap, **kw) if rank == 0 else None
    n = world * per
    ids = torch.arange(n, dtype=torch.int32, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(0)        # same frames on every rank
    for tick in range(cap // n + 2 * L):
      step = {
          'image': torch.randint(0, 255, (n, 64, 64, 3), dtype=torch.uint8, device='cuda', generator=gen),
          'reward': torch.randn(n, device='cuda', generator=gen),
          'is_first': (ids + tick) % 97 == 0,
          'is_last': (ids + tick) % 97 == 96,
          'env': ids,
          'tick': torch.full((n,), tick, dtype=torch.int32, device='cuda'),
      }
      shard.add_batch({k: v[rank * per:(rank + 1) * per] for k, v in step.items()})
      if single is not None:
        single.add_batch(step, list(range(n)))
    assert len(shard) == cap
    digests = []
    for _ in range(3):
      got = shard.sample(B)
      h = hashlib.sha256()
      for key in sorted(got):
        h.update(key.encode() + got[key].cpu().numpy().tobytes())
      want = None
      if single is not None:
        ref = single.sample(B)
        w = hashlib.sha256()
        for key in sorted(ref):
          w.update(key.encode() + ref[key].cpu().numpy().tobytes())
        want = w.hexdigest()
        owners = sorted(set((ref['env'][:, 0] // per).tolist()))
      digests.append((h.hexdigest(), want))
    out[rank] = (digests, owners if rank == 0 else None)
  finally:
    torch.distributed.destroy_process_group()


def test_crafter_config_256_envs_over_8_real_ranks():
  manager = mp.Manager()
  out = manager.dict()
  mp.spawn(_crafter_worker, args=(8, _free_port(), out), nprocs=8, join=True)
  assert sorted(out.keys()) == list(range(8))
  for i in range(3):
    want = out[0][0][i][1]
    assert want is not None
    for rank in range(8):
      assert out[rank][0][i][0] == want, (i, rank)     # every rank holds the single-replay batch
  assert len(out[0][1]) > 1                            # the last batch mixed sequences of several owners
