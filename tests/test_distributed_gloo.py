"""N>1 path on CPU: world_size-2 gloo process groups (127.0.0.1)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _worker(rank, world, port, fn, out):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from embodied_amd import distributed as D
  D.init('gloo')
  try:
    out[rank] = fn(rank, world, D)
  finally:
    torch.distributed.destroy_process_group()


def run_world(fn, world):
  manager = mp.Manager()
  for attempt in range(3):
    out = manager.dict()
    try:
      mp.spawn(_worker, args=(world, _free_port(), fn, out), nprocs=world, join=True)
      return dict(out)
    except Exception as e:
      # The port was free when _free_port looked and is taken again by the time the
      # store binds it (another process of the box): a new port, not a failed test.
      text = str(e)
      if attempt == 2 or not any(k in text for k in ('Address already in use', 'EADDRINUSE', 'errno: 98')):
        raise


def run2(fn):
  return run_world(fn, 2)


def _gather_job(rank, world, D):
  gen = np.random.default_rng(rank)
  batch = {
      'image': torch.as_tensor(gen.integers(0, 255, (3, 5, 4, 4, 2), dtype=np.uint8)),
      'reward': torch.as_tensor(gen.standard_normal((3, 5)).astype(np.float32)),
      'is_first': torch.as_tensor(gen.random((3, 5)) < 0.3),
      'stepid': torch.as_tensor(gen.integers(0, 255, (3, 5, 20), dtype=np.uint8)),
  }
  got = D.all_gather_batch(batch)
  layout = D.PackedLayout(
      [(k, v.dtype, v.shape[2:]) for k, v in batch.items()], 3, 5)
  flat = torch.zeros(layout.nbytes, dtype=torch.uint8)
  for k, v in layout.views(flat).items():
    v.copy_(batch[k])
  packed = D.all_gather_packed(flat, layout)
  grads = torch.full((1000,), float(rank + 1))
  D.all_reduce_mean(grads)
  return ({k: v.numpy() for k, v in got.items()},
          {k: v.numpy().copy() for k, v in packed.items()}, grads.numpy(),
          D.env_block(64).tolist(), D.max_over_ranks(rank + 0.5, 'cpu'))


def test_all_gather_and_all_reduce_world2():
  out = run2(_gather_job)
  for rank in (0, 1):
    got, packed, grads, block, mx = out[rank]
    for k in got:
      want = np.concatenate([
          _expected(r)[k] for r in (0, 1)], 0)
      assert np.array_equal(got[k], want), k
      assert np.array_equal(packed[k].reshape(want.shape), want), k
    assert np.allclose(grads, 1.5)
    assert block == list(range(rank * 32, rank * 32 + 32))
    assert mx == 1.5


def _expected(rank):
  gen = np.random.default_rng(rank)
  return {
      'image': gen.integers(0, 255, (3, 5, 4, 4, 2), dtype=np.uint8),
      'reward': gen.standard_normal((3, 5)).astype(np.float32),
      'is_first': gen.random((3, 5)) < 0.3,
      'stepid': gen.integers(0, 255, (3, 5, 20), dtype=np.uint8),
  }


def _index_job(rank, world, D):
  """Ranks that advance the host index identically draw identical row tables
  (what a replicated-index sharding relies on; host code only, no GPU)."""
  import ctypes as C
  from embodied_amd import _lib
  from embodied_amd._lib import api
  cfg = _lib.ReplayConfig(5, 40, 8, 64, 0, 0, 0, 1, 0)
  h = C.c_void_p()
  api.emb_replay_create(C.byref(cfg), None, 7, C.byref(h))
  workers = np.arange(6, dtype=np.int64)
  rows = np.zeros(6, np.int32)
  for _ in range(30):
    api.emb_replay_add_index(h, 6, _lib.ptr(workers), _lib.ptr(rows), None, None)
  table = np.zeros((4, 5), np.int32)
  api.emb_replay_sample_index(h, 4, 0, _lib.ptr(table), None, None)
  mine = torch.as_tensor(table)
  both = [torch.zeros_like(mine) for _ in range(world)]
  torch.distributed.all_gather(both, mine)
  api.raw.emb_replay_destroy(h)
  return bool((both[0] == both[1]).all()), table.tolist()


def test_replicated_index_draws_agree_world2():
  out = run2(_index_job)
  assert out[0][0] and out[1][0]
  assert out[0][1] == out[1][1]


def _comm_thread_job(rank, world, D):
  """The bench's pattern: async collectives issued from the comm thread, waited
  for one step later, drained before a collective on the calling thread."""
  comm = D.CommThread('cpu')
  grads = torch.zeros(257)
  pending, sums, gathers = [], [], []
  for step in range(6):
    for future in pending:
      future.result().wait()
    pending.clear()
    if step:
      sums.append(float(grads[0]))
      gathers.append(out.clone())
    grads.fill_(float((rank + 1) * (step + 1)))
    send = torch.full((5,), rank * 10 + step, dtype=torch.uint8)
    out = torch.empty(world * 5, dtype=torch.uint8)
    pending.append(comm.submit(lambda o=out, s=send: D.async_all_gather(o, s)))
    pending.append(comm.submit(lambda: D.async_all_reduce(grads)))
  for future in pending:
    future.result().wait()
  bad = comm.submit(lambda: 1 / 0)
  try:
    bad.result()
    raised = False
  except ZeroDivisionError:
    raised = True
  comm.close()
  final = D.max_over_ranks(float(rank), 'cpu')       # calling-thread collective after the drain
  return sums, [g.tolist() for g in gathers], raised, final


def test_comm_thread_issues_collectives_in_order_world2():
  out = run2(_comm_thread_job)
  for rank in (0, 1):
    sums, gathers, raised, final = out[rank]
    assert sums == [3.0 * (s + 1) for s in range(5)]
    assert gathers == [[s] * 5 + [10 + s] * 5 for s in range(5)]
    assert raised and final == 1.0


def test_all_gather_and_all_reduce_world4():
  """Four ranks: rank order in the gathered buffers, env blocks, the mean."""
  out = run_world(_gather_job, 4)
  for rank in range(4):
    got, packed, grads, block, mx = out[rank]
    for k in got:
      want = np.concatenate([_expected(r)[k] for r in range(4)], 0)
      assert np.array_equal(got[k], want), k
      assert np.array_equal(packed[k].reshape(want.shape), want), k
    assert np.allclose(grads, 2.5)
    assert block == list(range(rank * 16, rank * 16 + 16))
    assert mx == 3.5


def _slice_job(rank, world, D):
  """DP-slice exchange on host buffers: every rank builds a grouped packed
  batch (block d = what rank d gets) whose bytes name (source rank, block,
  sequence), exchanges, and returns what it received."""
  specs = [('image', torch.uint8, (4, 4, 2)), ('reward', torch.float32, ()),
           ('stepid', torch.uint8, (20,))]
  g, L = 2, 5
  layout = D.PackedLayout(specs, g, L)
  layout.groups = world
  flat = torch.zeros(world * layout.nbytes, dtype=torch.uint8)
  views = D.PackedViews(flat.view(world, layout.nbytes), layout)
  for d in range(world):
    for j in range(g):
      views['image'][d, j] = 40 * rank + 8 * d + j
      views['reward'][d, j] = 40 * rank + 8 * d + j + 0.5
      views['stepid'][d, j] = 40 * rank + 8 * d + j
  info = D.SampleInfo(layout, np.zeros(world * g, bool))
  work, out, got = D.exchange_dp_slices(flat, info)
  work.wait()
  return {k: got[k].numpy().copy() for k in ('image', 'reward', 'stepid')}


def test_dp_slice_exchange_world2_and_world4():
  for world in (2, 4):
    out = run_world(_slice_job, world)
    for rank in range(world):
      got = out[rank]
      assert got['image'].shape == (world, 2, 5, 4, 4, 2)
      for src in range(world):            # block `rank` of every source rank, in source order
        for j in range(2):
          tag = 40 * src + 8 * rank + j
          assert (got['image'][src, j] == tag).all()
          assert (got['reward'][src, j] == tag + 0.5).all()
          assert (got['stepid'][src, j] == tag).all()


def _clock_job(rank, world, D):
  """Three ranks whose own wall clocks are skewed (rank r sleeps r * 30 ms more
  per call): every decision must still come out the same on all of them."""
  import time
  always, never = D.GlobalClock(-1), D.GlobalClock(0)
  timed, far = D.GlobalClock(0.15), D.GlobalClock(1000)
  first = D.GlobalClock(0.15, first=True)
  decisions = []
  for call in range(12):
    time.sleep(0.02 + 0.03 * rank)
    decisions.append((
        always(call), never(call), timed(call), first(call), far(call),
        # rank 1 vetoes call 5 of the always-clock: nobody may act on it
        always(call, skip=(rank == 1 and call == 5))))
  return decisions


def test_global_clock_decides_alike_on_every_rank_world3():
  out = run_world(_clock_job, 3)
  assert out[0] == out[1] == out[2]
  always, never, timed, first, far, vetoed = zip(*out[0])
  assert always == (False,) + (True,) * 11           # the first call is skipped (clock.py:81-89)
  assert not any(never)
  assert not any(never) and not any(far)
  assert not timed[0] and any(timed) and any(first)   # rank 0's clock: due every 0.15 s
  assert vetoed[5] is False and all(vetoed[6:])


def test_global_clock_without_a_group_is_a_local_clock():
  from embodied_amd import distributed as D
  clock = D.GlobalClock(-1)
  assert not clock.multihost and clock(0) is True and clock(1, skip=True) is False
  assert D.GlobalClock(0)(0) is False


def _returns_of(rank, t, n=96):
  gen = np.random.default_rng(1000 * t + rank)
  return (gen.standard_normal(n) * (1 + rank) + 0.3 * t).astype(np.float32)


def _normalizer_job(rank, world, D):
  """The normalisers' collectives (embodied/jax/utils.py:76-88) as library calls:
  every rank feeds ITS returns, all ranks must end with the same statistics."""
  out = {}
  for impl in ('meanstd', 'perc', 'none'):
    norm = D.Normalize(impl, rate=0.05)
    for t in range(6):
      norm.update(torch.as_tensor(_returns_of(rank, t)))
    out[impl] = [float(v) for v in norm.stats(torch.zeros(()))]
  x = torch.as_tensor(_returns_of(rank, 0))
  out['pmean_of_means'] = D.pmean(torch.stack([x.mean(), x.square().mean()])).numpy()
  out['percentiles'] = D.percentile_over_ranks(x, [5.0, 50.0, 95.0]).numpy()
  return out


def test_normalizer_collectives_match_the_single_process_statement():
  """pmean of local means and percentiles of the all-gathered returns, with 2 and
  3 gloo ranks, against numpy on the concatenation and against the oracle's
  single-process Normalize fed every rank's values."""
  from oracle import np_oracle
  for world in (2, 3):
    got = run_world(_normalizer_job, world)
    parts0 = [_returns_of(r, 0) for r in range(world)]
    want_perc = np.percentile(np.concatenate(parts0), [5.0, 50.0, 95.0])
    want_mean = [np.mean([p.mean() for p in parts0]), np.mean([np.square(p).mean() for p in parts0])]
    for r in range(world):
      np.testing.assert_allclose(got[r]['percentiles'], want_perc, rtol=1e-6, atol=1e-6)
      np.testing.assert_allclose(got[r]['pmean_of_means'], want_mean, rtol=1e-6, atol=1e-6)
      assert got[r]['none'] == [0.0, 1.0]
    for impl in ('meanstd', 'perc'):
      ref = np_oracle.Normalize(impl, rate=0.05)
      for t in range(6):
        ref.update([_returns_of(r, t) for r in range(world)])
      want = [float(v) for v in ref.stats()]
      for r in range(world):
        np.testing.assert_allclose(got[r][impl], want, rtol=1e-5, atol=1e-6)
        assert got[r][impl] == got[0][impl]            # every rank holds the same statistics


def _exchange_contract_job(rank, world, D):
  """A train step's collectives through the exchange / wait contract that
  bench.py's timed path uses (NativeComm on RCCL, GroupComm on a process group):
  issue now, complete one train step later, buffers untouched in between."""
  comm = D.GroupComm()
  block, steps = 64, 5
  sent, recv, grads, log = [], [], [], []
  for k in range(steps):
    # block d of rank r's batch k carries the byte (16 * r + 4 * k + d) & 0xFF
    flat = torch.cat([torch.full((block,), (16 * rank + 4 * k + d) & 0xFF, dtype=torch.uint8)
                      for d in range(world)])
    sent.append(flat)
    recv.append(torch.zeros_like(flat))
    grads.append(torch.full((1000,), float(rank + 10 * k)))
    comm.wait()                                   # completes exchange k-1 (a no-op at k = 0)
    if k:
      log.append((recv[k - 1].clone().numpy(), grads[k - 1].clone().numpy()))
    if k % 2 == 0:
      comm.exchange(sent[k], recv[k], grads[k])   # slices + gradients in one exchange
    else:
      comm.exchange(grads=grads[k])               # a batch without fresh windows: gradients only
  try:
    comm.exchange(grads=grads[0])
    double = False
  except AssertionError:
    double = True                                 # a second exchange before wait() is refused
  comm.wait()
  log.append((recv[-1].numpy(), grads[-1].numpy()))
  return log, double


def test_exchange_wait_contract_with_two_and_three_ranks():
  for world in (2, 3):
    got = run_world(_exchange_contract_job, world)
    for rank in range(world):
      log, double = got[rank]
      assert double and len(log) == 5
      for k, (received, grads) in enumerate(log):
        np.testing.assert_allclose(grads, np.mean([r + 10 * k for r in range(world)]))
        want = np.zeros(64 * world, np.uint8)
        if k % 2 == 0:        # block s came from rank s: its block `rank` of batch k
          want = np.concatenate([np.full(64, (16 * s + 4 * k + rank) & 0xFF, np.uint8)
                                 for s in range(world)])
        assert (received == want).all(), (world, rank, k)


def _subgroup_job(rank, world, D):
  """GroupComm on a sub-group: its collectives must run on THAT communicator
  (two of three ranks exchange while the third does nothing)."""
  import torch.distributed as dist
  pair = dist.new_group([0, 1])                  # every rank makes the group
  if rank == 2:
    return None
  comm = D.GroupComm(pair)
  assert comm.world == 2
  flat = torch.cat([torch.full((32,), 10 * rank + d, dtype=torch.uint8) for d in range(2)])
  received = torch.zeros_like(flat)
  grads = torch.full((100,), float(rank + 1))
  comm.exchange(flat, received, grads)
  comm.wait()
  return received.numpy(), grads.numpy()


def test_group_comm_uses_its_own_group_world3():
  got = run_world(_subgroup_job, 3)
  assert got[2] is None
  for rank in (0, 1):
    received, grads = got[rank]
    want = np.concatenate([np.full(32, 10 * s + rank, np.uint8) for s in range(2)])
    assert (received == want).all()
    np.testing.assert_allclose(grads, 1.5)       # mean over the group's two ranks, not over three


def test_normalize_stats_before_any_update():
  from embodied_amd import distributed as D
  assert D.Normalize('none').stats() == (0.0, 1.0)
  lo, scale = D.Normalize('perc').stats()
  assert float(lo) == 0.0 and float(scale) == float(np.float32(1e-8))       # zero statistics, the limit as scale
  mean, std = D.Normalize('meanstd', debias=False).stats(torch.zeros(()))
  assert float(mean) == 0.0 and float(std) == float(np.float32(1e-8))


def _pretrain_job(rank, world, D):
  """`run.pretrain` on every rank of a job (embodied/run/pretrain.py:8-96 with
  `replicas` > 1): the report / log / save decisions come from GlobalClocks, so
  every rank takes them at the same steps; only replica 0 writes checkpoints."""
  import tempfile
  import time
  import types
  import embodied_amd as emb
  import shutil
  logdir = tempfile.mkdtemp(prefix=f'emb_pretrain_{rank}_')      # (a fresh one: a checkpoint left behind would be resumed)

  class Stream:
    def __init__(self):
      self.k = 0
    def __iter__(self):
      return self
    def __next__(self):
      self.k += 1
      return {'count': np.arange(self.k, self.k + 2)}

  class Model:
    def __init__(self):
      self.trains, self.report_steps, self.saves = 0, [], 0
    def stream(self, st):
      return st
    def init_train(self, n):
      return ()
    init_report = init_train
    def train(self, carry, batch):
      self.trains += 1
      time.sleep(0.002 * (1 + rank))            # ranks of different speed: rank 0's clock decides
      return carry, {}, {'loss': 1.0}
    def report(self, carry, batch):
      self.report_steps.append(self.trains)
      return carry, {'m': 1.0}
    def save(self):
      self.saves += 1
      return {}
    def load(self, data):
      pass

  model = Model()
  args = types.SimpleNamespace(
      logdir=logdir, steps=60, batch_size=2, batch_length=4, log_every=0.03, report_every=0.05,
      save_every=0.04, consec_report=1, report_batches=1, replica=rank, from_checkpoint='')
  emb.run.pretrain(lambda: model, lambda replay, mode: Stream(), lambda: emb.utils.Logger(), args)
  wrote = os.path.exists(os.path.join(logdir, 'checkpoint.pkl'))
  shutil.rmtree(logdir, ignore_errors=True)
  return model.trains, sorted(set(model.report_steps)), model.saves, wrote


def test_pretrain_takes_its_decisions_in_lockstep_world2():
  out = run2(_pretrain_job)
  (trains0, reports0, saves0, wrote0), (trains1, reports1, saves1, wrote1) = out[0], out[1]
  assert trains0 == trains1 == 60
  assert reports0 == reports1 and len(reports0) >= 1      # the same train steps on both ranks
  assert wrote0 and not wrote1 and saves0 >= 1 and saves1 == 0
