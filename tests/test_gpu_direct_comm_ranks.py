"""The direct xGMI schedule (`emb_direct_*`, csrc/direct_comm.hip): the gradient
all-reduce (embodied/jax/opt.py:52-54) as reduce-scatter + all-gather and the
DP-slice all-to-all (embodied/jax/internal.py:145-152), every rank writing its
peers' shares through hipIpc pointers, all peers at once.

Checked with 2 and 3 REAL ranks (processes) that share the test box's one GPU --
the peers' memory is reached through the same hipIpc handles as on a node with a
GPU per rank, only the stores stay on one device -- against numpy on the
concatenation of every rank's seeded inputs, exactly like the loopback test of
the RCCL entry points (tests/test_gpu_native_comm_ranks.py).  Every in-kernel
wait is bounded: a rank that never arrives is an error word, not a hung GPU.

With at least two GPUs visible the same worker also runs with one GPU per rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SLICE_BYTES = 700_016                  # 16-byte multiple; + an odd size below
GRAD_NUMEL = 300_007                   # not a multiple of world * 8: a short last shard and a scalar tail


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _bytes_of(rank, tag, n):
  return np.random.default_rng([17, tag, rank]).integers(0, 256, n, dtype=np.uint8)


def _floats_of(rank, tag, n):
  return np.random.default_rng([23, tag, rank]).standard_normal(n).astype(np.float32)


def _ints_of(rank, tag, n):
  return np.random.default_rng([29, tag, rank]).integers(-8, 9, n).astype(np.float32)


DTYPES = (('f32', torch.float32), ('bf16', torch.bfloat16), ('f16', torch.float16))


def _worker(rank, world, port, out, per_gpu):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if per_gpu else 0),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from embodied_amd import distributed as D
  torch.cuda.set_device(rank if per_gpu else 0)
  D.init('gloo')
  try:
    comm = D.DirectComm(rank, world, max_grad_bytes=4 << 20, max_slice_bytes=1 << 20, timeout_ms=15000)
    cuda = lambda a: torch.as_tensor(a).cuda()
    res = {}
    res['all_to_all'] = comm.all_to_all(cuda(_bytes_of(rank, 1, world * SLICE_BYTES))).cpu().numpy()
    res['all_to_all_odd'] = comm.all_to_all(cuda(_bytes_of(rank, 6, world * 1001))).cpu().numpy()
    # the trajectory all-gather (north_star; embodied/jax/internal.py:145-152): ONE block per rank to every peer
    res['all_gather'] = comm.all_gather(cuda(_bytes_of(rank, 8, SLICE_BYTES))).cpu().numpy()
    res['all_gather_odd'] = comm.all_gather(cuda(_bytes_of(rank, 9, 1001))).cpu().numpy()
    for name, dtype in DTYPES:
      whole = cuda(_ints_of(rank, 2, GRAD_NUMEL)).to(dtype)
      comm.all_reduce(whole, mean=False)
      res[f'sum_{name}'] = whole.double().cpu().numpy()
      noisy = cuda(_floats_of(rank, 3, GRAD_NUMEL)).to(dtype)
      comm.all_reduce(noisy, mean=True)
      res[f'mean_{name}'] = noisy.double().cpu().numpy()
    tiny = cuda(_floats_of(rank, 7, 5))               # fewer elements than ranks * vector width
    comm.all_reduce(tiny, mean=False)
    res['tiny'] = tiny.cpu().numpy()
    # Many operations back to back (the two slot sets alternate, flags count up):
    # the buffers are still being produced on the stream when exchange() is called.
    group = D.GroupComm()
    log = []
    keep = []
    for k in range(6):
      src = cuda(_bytes_of(rank, 10 + k, world * SLICE_BYTES))
      slices = torch.empty_like(src)
      slices.copy_(src)
      slices.bitwise_xor_(0x5A)
      grads = cuda(_floats_of(rank, 20 + k, GRAD_NUMEL)).to(torch.bfloat16)
      grads.mul_(2)
      twin_slices, twin_grads = slices.clone(), grads.clone()
      received, twin_received = torch.zeros_like(slices), torch.zeros_like(slices)
      comm.wait()
      if k % 2 == 0:
        comm.exchange(slices, received, grads)
        group.exchange(twin_slices, twin_received, twin_grads)
      else:
        comm.exchange(grads=grads)
        group.exchange(grads=twin_grads)
      group.wait()
      keep.append((slices, received, grads))
      log.append((received, grads, twin_received, twin_grads))
    comm.wait()
    res['exchange'] = [(a.cpu().numpy(), b.float().cpu().numpy(), c.cpu().numpy(), d.float().cpu().numpy())
                       for a, b, c, d in log]
    # exchange(gather=True): the trajectory all-gather + the f32 gradient all-reduce in one call
    gathers = []
    for k in range(4):
      mine = cuda(_bytes_of(rank, 40 + k, SLICE_BYTES if k % 2 == 0 else 1001))
      grads = cuda(_floats_of(rank, 50 + k, GRAD_NUMEL))
      everyone = torch.zeros(world * mine.numel(), dtype=torch.uint8, device='cuda')
      twin_everyone, twin_grads = torch.zeros_like(everyone), grads.clone()
      comm.wait()
      comm.exchange(mine, everyone, grads, gather=True)
      group.exchange(mine.clone(), twin_everyone, twin_grads, gather=True)
      group.wait()
      keep.append((mine, everyone, grads))
      gathers.append((everyone, grads, twin_everyone, twin_grads))
    comm.wait()
    res['gathers'] = [(a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy(), d.cpu().numpy()) for a, b, c, d in gathers]
    res['timed_out'] = comm.timed_out()
    try:
      comm.all_reduce(torch.zeros(8, dtype=torch.float64, device='cuda'))
      res['refused'] = False
    except AssertionError:
      res['refused'] = True
    try:      # a view that does not start on a 16-byte boundary: refused, not reduced with misaligned 16-byte accesses
      comm.all_reduce(torch.zeros(4099, device='cuda')[1:])
      res['refused_unaligned'] = False
    except AssertionError:
      res['refused_unaligned'] = True
    comm.close()
    out[rank] = res
  finally:
    torch.distributed.destroy_process_group()


def _check(out, world):
  assert sorted(out.keys()) == list(range(world))
  ranks = range(world)
  sums = np.sum([_ints_of(r, 2, GRAD_NUMEL).astype(np.float64) for r in ranks], 0)
  for rank in ranks:
    got = out[rank]
    assert got['timed_out'] is False                         # nobody gave up on a peer
    for key, tag, size in (('all_to_all', 1, SLICE_BYTES), ('all_to_all_odd', 6, 1001)):
      want = np.concatenate([_bytes_of(s, tag, world * size)[rank * size:(rank + 1) * size] for s in ranks])
      assert np.array_equal(got[key], want), key
    for key, tag, size in (('all_gather', 8, SLICE_BYTES), ('all_gather_odd', 9, 1001)):
      assert np.array_equal(got[key], np.concatenate([_bytes_of(s, tag, size) for s in ranks])), key
    for name, tol in (('f32', 1e-6), ('bf16', 2e-2), ('f16', 2e-3)):
      assert np.array_equal(got[f'sum_{name}'], sums), name  # small integers: exact in every dtype
      parts = [torch.as_tensor(_floats_of(r, 3, GRAD_NUMEL)).to(dict(DTYPES)[name]).double().numpy()
               for r in ranks]
      np.testing.assert_allclose(got[f'mean_{name}'], np.mean(parts, 0), rtol=tol, atol=tol)
      assert np.array_equal(got[f'mean_{name}'], out[0][f'mean_{name}'])      # all ranks alike, to the bit
    np.testing.assert_allclose(got['tiny'], np.sum([_floats_of(r, 7, 5) for r in ranks], 0), rtol=1e-6)
    for k, (received, grads, twin_received, twin_grads) in enumerate(got['exchange']):
      want_grads = np.mean([
          (torch.as_tensor(_floats_of(r, 20 + k, GRAD_NUMEL)).to(torch.bfloat16) * 2).float().numpy()
          for r in ranks], 0)
      np.testing.assert_allclose(grads, want_grads, rtol=2e-2, atol=2e-2)
      np.testing.assert_allclose(grads, twin_grads, rtol=2e-2, atol=2e-2)    # == GroupComm on gloo
      assert np.array_equal(grads, out[0]['exchange'][k][1])                  # all ranks alike, to the bit
      want = np.zeros(world * SLICE_BYTES, np.uint8)
      if k % 2 == 0:
        want = np.concatenate([
            (_bytes_of(s, 10 + k, world * SLICE_BYTES) ^ 0x5A)[rank * SLICE_BYTES:(rank + 1) * SLICE_BYTES]
            for s in ranks])
      assert np.array_equal(received, want), (rank, k)
      assert np.array_equal(received, twin_received), (rank, k)
    for k, (everyone, grads, twin_everyone, twin_grads) in enumerate(got['gathers']):
      size = SLICE_BYTES if k % 2 == 0 else 1001
      assert np.array_equal(everyone, np.concatenate([_bytes_of(s, 40 + k, size) for s in ranks])), (rank, k)
      assert np.array_equal(everyone, twin_everyone), (rank, k)
      np.testing.assert_allclose(grads, np.mean([_floats_of(r, 50 + k, GRAD_NUMEL) for r in ranks], 0),
                                 rtol=1e-5, atol=1e-6)
      np.testing.assert_allclose(grads, twin_grads, rtol=1e-5, atol=1e-6)
      assert np.array_equal(grads, out[0]['gathers'][k][1])                   # all ranks alike, to the bit
    assert got['refused'] is True and got['refused_unaligned'] is True


@pytest.mark.parametrize('world', [2, 3])
def test_direct_schedule_between_real_ranks_on_one_gpu(world):
  manager = mp.Manager()
  out = manager.dict()
  mp.spawn(_worker, args=(world, _free_port(), out, False), nprocs=world, join=True)
  _check(out, world)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs one GPU per rank')
def test_direct_schedule_with_one_gpu_per_rank():
  """Switches itself on wherever the suite sees at least two GPUs: the peers'
  memory is then on another device, behind an xGMI link."""
  world = min(torch.cuda.device_count(), 8)
  manager = mp.Manager()
  out = manager.dict()
  mp.spawn(_worker, args=(world, _free_port(), out, True), nprocs=world, join=True)
  _check(out, world)


def test_world_of_one_is_the_identity():
  from embodied_amd import distributed as D
  comm = D.DirectComm(0, 1, max_grad_bytes=1 << 20, max_slice_bytes=1 << 20)
  x = torch.arange(1000, dtype=torch.float32, device='cuda')
  assert torch.equal(comm.all_reduce(x.clone(), mean=True), x)
  flat = torch.randint(0, 255, (4096,), dtype=torch.uint8, device='cuda')
  assert torch.equal(comm.all_to_all(flat), flat)
  assert comm.timed_out() is False
  comm.close()


def _lonely_worker(rank, world, port, out):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  import time
  from embodied_amd import distributed as D
  torch.cuda.set_device(0)
  D.init('gloo')
  try:
    comm = D.DirectComm(rank, world, max_grad_bytes=1 << 20, max_slice_bytes=1 << 20, timeout_ms=300)
    res = {}
    if rank == 0:
      grads = torch.arange(4096, dtype=torch.float32, device='cuda')
      began = time.perf_counter()
      comm.all_reduce(grads)                 # rank 1 never joins this one
      torch.cuda.synchronize()
      res['seconds'] = time.perf_counter() - began
      res['timed_out'] = comm.timed_out()
      # fatal, not silent: nothing was reduced into the buffer ...
      res['untouched'] = bool(torch.equal(grads, torch.arange(4096, dtype=torch.float32, device='cuda')))
      # ... and every later call on the dead communicator raises
      raised = []
      for call in (lambda: comm.wait(), lambda: comm.exchange(grads=grads), lambda: comm.all_reduce(grads),
                   lambda: comm.all_to_all(torch.zeros(64, dtype=torch.uint8, device='cuda')), comm.check):
        try:
          call()
          raised.append(None)
        except RuntimeError as e:          # (EmbError: a transport failure, not a bad argument)
          raised.append(str(e))
      res['raised'] = raised
    torch.distributed.barrier()              # rank 1 stays alive (its memory mapped) until rank 0 is through
    if rank == 1:
      # The dead rank's reduce raised no flag for the operation it gave up on: its peer's
      # collect runs into the same time-out (what its buffer holds then is undefined -- its
      # own shard is reduced, the dead rank's never arrives -- and it is told so).
      grads = torch.ones(4096, device='cuda')
      comm.all_reduce(grads)
      torch.cuda.synchronize()
      res['timed_out'] = comm.timed_out()
      try:
        comm.wait()
        res['raised'] = None
      except RuntimeError as e:
        res['raised'] = str(e)
    out[rank] = res
    torch.distributed.barrier()
    comm.close()
  finally:
    torch.distributed.destroy_process_group()


def test_a_peer_that_never_arrives_kills_the_communicator_not_the_gpu():
  """Every wait inside the kernels is bounded by `timeout_ms` -- and giving up is
  fatal for the communicator, never a silently wrong gradient: the kernel that
  gave up writes no result and raises no flag, `timed_out()` says what happened,
  every later call raises, and the late peer runs into the same time-out."""
  manager = mp.Manager()
  out = manager.dict()
  mp.spawn(_lonely_worker, args=(2, _free_port(), out), nprocs=2, join=True)
  first, late = out[0], out[1]
  assert first['timed_out'] is True and first['untouched'] is True
  assert 0.25 < first['seconds'] < 5.0        # two bounded waits (reduce, collect) of at most 0.3 s each
  assert all(msg and 'timed out' in msg for msg in first['raised']), first['raised']
  assert late['timed_out'] is True and 'timed out' in late['raised']


def _failing_worker(rank, world, port, out):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from embodied_amd import distributed as D
  torch.cuda.set_device(0)
  D.init('gloo')
  try:
    # rank 1 asks for a buffer no GPU has: its setup fails, rank 0's is fine
    size = (1 << 50) if rank == 1 else (1 << 20)
    try:
      D.DirectComm(rank, world, max_grad_bytes=size, max_slice_bytes=1 << 20, timeout_ms=300)
      out[rank] = 'constructed'
    except RuntimeError as e:
      out[rank] = str(e)
    torch.distributed.barrier()
  finally:
    torch.distributed.destroy_process_group()


def test_a_rank_that_cannot_set_up_fails_every_rank_together():
  """Setup takes part in both of its rounds whatever happened locally: the
  healthy rank raises too (naming the rank that failed) instead of waiting for
  a handle that never comes -- bench.py's `--comm auto` then stays on RCCL."""
  manager = mp.Manager()
  out = manager.dict()
  mp.spawn(_failing_worker, args=(2, _free_port(), out), nprocs=2, join=True)
  assert 'rank(s) [1] could not set up' in out[0], out[0]
  assert 'rank(s) [1] could not set up' in out[1] and 'here:' in out[1], out[1]


STRESS_SIZES = (1, 7, 4096, 300_007, 1 << 20)            # elements; also bytes per all-to-all block


def _stress_worker(rank, world, port, out, per_gpu, rounds):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if per_gpu else 0),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from embodied_amd import distributed as D
  torch.cuda.set_device(rank if per_gpu else 0)
  D.init('gloo')
  try:
    comm = D.DirectComm(rank, world, max_grad_bytes=4 << 20, max_slice_bytes=1 << 20, timeout_ms=20000)
    bad = torch.zeros(3, dtype=torch.int64, device='cuda')       # all-reduce, all-to-all, exchange
    peers = torch.arange(world, device='cuda')
    for i in range(rounds):
      n = STRESS_SIZES[i % len(STRESS_SIZES)]
      # every operation's payload is a function of (rank, i): a slot read one
      # operation late, or overwritten one early, shows as a wrong value
      grads = torch.full((n,), float(rank + 1 + i % 7), device='cuda')
      want = world * (world + 1) / 2 + world * (i % 7)
      blocks = ((peers * 16 + rank * 3 + i) % 251).to(torch.uint8)              # block p of mine
      flat = blocks.repeat_interleave(n)
      arrive = ((rank * 16 + peers * 3 + i) % 251).to(torch.uint8).repeat_interleave(n)   # block s from rank s
      if i % 3 == 2:
        received = torch.empty_like(flat)
        comm.wait()
        comm.exchange(flat, received, grads, mean=False)
        comm.wait()
        bad[2] += (grads != want).any() | (received != arrive).any()
      else:
        comm.all_reduce(grads, mean=False)
        bad[0] += (grads != want).any()
        bad[1] += (comm.all_to_all(flat) != arrive).any()
    torch.cuda.synchronize()
    out[rank] = {'bad': bad.cpu().tolist(), 'timed_out': comm.timed_out()}
    torch.distributed.barrier()
    comm.close()
  finally:
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_a_thousand_operations_back_to_back(world):
  """The two slot sets alternate and the sequence flags count up through 1 000
  operations of five sizes, queued without a host synchronisation in between:
  every result is checked on the device against the value its (rank, round)
  alone determines."""
  manager = mp.Manager()
  out = manager.dict()
  mp.spawn(_stress_worker, args=(world, _free_port(), out, False, 1000), nprocs=world, join=True)
  for rank in range(world):
    assert out[rank] == {'bad': [0, 0, 0], 'timed_out': False}, (rank, out[rank])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs one GPU per rank')
def test_a_thousand_operations_back_to_back_with_one_gpu_per_rank():
  world = min(torch.cuda.device_count(), 8)
  manager = mp.Manager()
  out = manager.dict()
  mp.spawn(_stress_worker, args=(world, _free_port(), out, True, 1000), nprocs=world, join=True)
  for rank in range(world):
    assert out[rank] == {'bad': [0, 0, 0], 'timed_out': False}, (rank, out[rank])
