"""The library's own collective entry points (`emb_comm_*`, include/embodied_hip.h;
the call sites they replace: embodied/jax/opt.py:52-54 gradient pmean,
embodied/jax/utils.py:76-88 normaliser pmean / all-gather,
embodied/jax/internal.py:145-152 per-replica -> global batch) with 2 and 3 REAL
ranks on the test box's one GPU.

RCCL refuses two ranks on one device, so the ranks bind the ten RCCL symbols from
tests/fake_rccl/libfake_rccl.so instead (EMB_RCCL_LIB, abi.cpp rccl()): a
loopback transport between processes through shared memory, stream-ordered like
the real thing.  Everything above that table -- emb_comm_init / exchange / wait /
allreduce_grads_as / alltoall_slices / allgather_* / pmean_scalars, their
offsets, counts, dtype codes, group fusing and stream forks -- is the product
code, checked against numpy on the concatenation of every rank's seeded inputs
and against `GroupComm` on a gloo process group."""
import ctypes
import importlib.util
import os
import pathlib
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = pathlib.Path(__file__).resolve().parent


def fake_rccl():
  spec = importlib.util.spec_from_file_location('_fake_rccl_build', HERE / 'fake_rccl' / 'build.py')
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return str(mod.build())


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


GATHER_BYTES = (3 << 20) + 17          # > one 1 MiB staging slot: the chunk loop runs
SLICE_BYTES = 700_016                  # > slot / world: an all-to-all of several rounds
GRAD_NUMEL = 300_000                   # f32: 1.2 MB, two slot loads


def _worker(rank, world, port, lib, out):
  """lib = the loopback stand-in (all ranks on GPU 0), or None: REAL RCCL, one GPU per rank."""
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0' if lib else str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  if lib:
    os.environ.update(EMB_RCCL_LIB=lib, FAKE_RCCL_SLOT_BYTES=str(1 << 20), FAKE_RCCL_TIMEOUT_S='45')
  else:
    os.environ.pop('EMB_RCCL_LIB', None)
  from embodied_amd import distributed as D
  from tests.test_distributed_gloo import _returns_of
  torch.cuda.set_device(0 if lib else rank)
  D.init('gloo')
  try:
    comm = D.NativeComm(rank, world)               # id from rank 0 through the gloo group
    cuda = lambda a: torch.as_tensor(a).cuda()
    res = {}
    res['all_gather'] = comm.all_gather(cuda(_bytes_of(rank, 0, GATHER_BYTES))).cpu().numpy()
    res['all_to_all'] = comm.all_to_all(cuda(_bytes_of(rank, 1, world * SLICE_BYTES))).cpu().numpy()
    for name, dtype in (('f32', torch.float32), ('bf16', torch.bfloat16), ('f16', torch.float16),
                        ('f64', torch.float64)):
      whole = cuda(_ints_of(rank, 2, GRAD_NUMEL)).to(dtype)
      comm.all_reduce(whole, mean=False)
      res[f'sum_{name}'] = whole.double().cpu().numpy()
      noisy = cuda(_floats_of(rank, 3, GRAD_NUMEL)).to(dtype)
      comm.all_reduce(noisy, mean=True)
      res[f'mean_{name}'] = noisy.double().cpu().numpy()
    res['returns'] = comm.all_gather_returns(cuda(_floats_of(rank, 4, 16 * 63))).cpu().numpy()
    res['pmean'] = comm.pmean(cuda(_floats_of(rank, 5, 7))).cpu().numpy()

    # One train step's exchange: slices + gradients on the communicator's own
    # stream, AFTER what the current stream has queued (the buffers are still
    # being produced by kernels when exchange() is called), completed by wait()
    # one train step later; gradient-only exchanges in between, as in bench.py.
    group = D.GroupComm()
    log = []
    keep = []
    for k in range(4):
      src = cuda(_bytes_of(rank, 10 + k, world * SLICE_BYTES))
      slices = torch.empty_like(src)
      slices.copy_(src)
      slices.bitwise_xor_(0x5A)                     # produced on the stream, no sync
      grads = cuda(_floats_of(rank, 20 + k, GRAD_NUMEL)).to(torch.bfloat16)
      grads.mul_(2)
      twin_slices, twin_grads = slices.clone(), grads.clone()
      received, twin_received = torch.zeros_like(slices), torch.zeros_like(slices)
      comm.wait()                                   # exchange k-1 (a no-op at k = 0)
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
    comm.wait()                                     # nothing in flight: a no-op
    # exchange(gather=True): the trajectory all-gather north_star names + the f32 gradient
    # all-reduce in ONE call on the communicator's stream (emb_comm_exchange_gather, ABI 5)
    mine = cuda(_bytes_of(rank, 40, SLICE_BYTES + 3))
    everyone = torch.zeros(world * mine.numel(), dtype=torch.uint8, device='cuda')
    g32 = cuda(_floats_of(rank, 41, GRAD_NUMEL))
    comm.exchange(mine, everyone, g32, gather=True)
    comm.wait()
    res['gather_exchange'] = (everyone.cpu().numpy(), g32.cpu().numpy())
    res['exchange'] = [(a.cpu().numpy(), b.float().cpu().numpy(), c.cpu().numpy(), d.float().cpu().numpy())
                       for a, b, c, d in log]

    # The normalisers on the native communicator (Normalize._mean's pmean and the
    # 'perc' all-gather) -- every rank feeds ITS returns.
    for impl in ('meanstd', 'perc'):
      norm = D.Normalize(impl, rate=0.05, comm=comm)
      for t in range(6):
        norm.update(cuda(_returns_of(rank, t)))
      res[impl] = [float(v) for v in norm.stats()]
    res['percentiles'] = D.percentile_over_ranks(
        cuda(_returns_of(rank, 0)), [5.0, 50.0, 95.0], comm=comm).cpu().numpy()

    # dtype codes the gradient all-reduce does not take are refused before any rank blocks
    try:
      comm._api.emb_comm_allreduce_grads_as(
          comm._handle, received.data_ptr(), 16, comm._lib.U8, 0, comm._lib.raw_stream(received.device))
      res['refused'] = False
    except Exception as e:
      res['refused'] = 'dtype' in str(e)
    res['ops'] = int(ctypes.CDLL(lib).fake_rccl_ops_done()) if lib else 1000
    comm.close()
    out[rank] = res
  finally:
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_native_collectives_between_real_ranks(world):
  _run_and_check(world, fake_rccl())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='real RCCL needs one GPU per rank')
def test_native_collectives_on_real_rccl():
  """Switches itself on wherever the suite sees at least two GPUs: the same
  worker, the same checks, the ten symbols bound from the real librccl (no
  EMB_RCCL_LIB), one GPU per rank -- RCCL's own init ordering, group semantics
  and stream behaviour at N > 1."""
  _run_and_check(min(torch.cuda.device_count(), 4), None)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='real RCCL needs one GPU per rank')
def test_bench_two_ranks_on_real_rccl():
  """`python bench.py --gpus 2` as the driver runs it: nccl process group, the
  library's own RCCL exchange in the timed path after its self-check against
  torch.distributed passed on both ranks."""
  from tests.test_gpu_bench_launcher import run_bench
  rec = run_bench('--gpus', '2', '--steps', '20', '--warmup', '5')
  assert rec['n_gpus'] == 2 and rec['rccl_ranks'] == 2 and rec['backend'] == 'nccl'
  native = rec['native_comm']
  assert native['status'] == 'ok' and native['ranks'] == 2 and all(native['checks'].values()), native
  assert native['transport'] == 'rccl' and native['timed_path'] == native['auto']['chose']
  assert ('emb_direct_exchange' if native['timed_path'] == 'direct' else 'emb_comm_exchange') in rec['config']['parallelism']
  assert rec['value'] > 0 and rec['replicas_only']['env_steps_per_s'] > 0
  direct = native.get('direct')                       # the direct xGMI schedule, timed beside RCCL
  assert direct and direct['status'] == 'ok', direct


def _run_and_check(world, lib):
  manager = mp.Manager()
  out = manager.dict()
  mp.spawn(_worker, args=(world, _free_port(), lib, out), nprocs=world, join=True)
  assert sorted(out.keys()) == list(range(world))
  from oracle import np_oracle
  from tests.test_distributed_gloo import _returns_of
  ranks = range(world)
  want_gather = np.concatenate([_bytes_of(r, 0, GATHER_BYTES) for r in ranks])
  sums = np.sum([_ints_of(r, 2, GRAD_NUMEL).astype(np.float64) for r in ranks], 0)
  for rank in ranks:
    got = out[rank]
    assert got['ops'] > 30                                   # the loopback transport carried them
    assert np.array_equal(got['all_gather'], want_gather)
    # block s of the result came from rank s: its block `rank`
    want = np.concatenate([_bytes_of(s, 1, world * SLICE_BYTES)[rank * SLICE_BYTES:(rank + 1) * SLICE_BYTES]
                           for s in ranks])
    assert np.array_equal(got['all_to_all'], want)
    for name, tol in (('f32', 1e-6), ('bf16', 2e-2), ('f16', 2e-3), ('f64', 1e-12)):
      assert np.array_equal(got[f'sum_{name}'], sums), name  # small integers: exact in every dtype
      parts = [torch.as_tensor(_floats_of(r, 3, GRAD_NUMEL)).to(
          {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16, 'f64': torch.float64}[name]
      ).double().numpy() for r in ranks]
      np.testing.assert_allclose(got[f'mean_{name}'], np.mean(parts, 0), rtol=tol, atol=tol)
      assert np.array_equal(got[f'mean_{name}'], out[0][f'mean_{name}'])      # all ranks alike, to the bit
    assert np.array_equal(got['returns'], np.concatenate([_floats_of(r, 4, 16 * 63) for r in ranks]))
    np.testing.assert_allclose(got['pmean'], np.mean([_floats_of(r, 5, 7) for r in ranks], 0),
                               rtol=1e-6, atol=1e-7)
    everyone, g32 = got['gather_exchange']
    assert np.array_equal(everyone, np.concatenate([_bytes_of(s, 40, SLICE_BYTES + 3) for s in ranks]))
    np.testing.assert_allclose(g32, np.mean([_floats_of(r, 41, GRAD_NUMEL) for r in ranks], 0), rtol=1e-5, atol=1e-6)
    for k, (received, grads, twin_received, twin_grads) in enumerate(got['exchange']):
      want_grads = np.mean([
          (torch.as_tensor(_floats_of(r, 20 + k, GRAD_NUMEL)).to(torch.bfloat16) * 2).float().numpy()
          for r in ranks], 0)
      np.testing.assert_allclose(grads, want_grads, rtol=2e-2, atol=2e-2)
      np.testing.assert_allclose(grads, twin_grads, rtol=2e-2, atol=2e-2)    # == GroupComm on gloo
      want = np.zeros(world * SLICE_BYTES, np.uint8)
      if k % 2 == 0:
        want = np.concatenate([
            (_bytes_of(s, 10 + k, world * SLICE_BYTES) ^ 0x5A)[rank * SLICE_BYTES:(rank + 1) * SLICE_BYTES]
            for s in ranks])
      assert np.array_equal(received, want), (rank, k)
      assert np.array_equal(received, twin_received), (rank, k)
    parts0 = [_returns_of(r, 0) for r in ranks]
    np.testing.assert_allclose(got['percentiles'], np.percentile(np.concatenate(parts0), [5.0, 50.0, 95.0]),
                               rtol=1e-6, atol=1e-6)
    for impl in ('meanstd', 'perc'):
      ref = np_oracle.Normalize(impl, rate=0.05)
      for t in range(6):
        ref.update([_returns_of(r, t) for r in ranks])
      np.testing.assert_allclose(got[impl], [float(v) for v in ref.stats()], rtol=1e-5, atol=1e-6)
      assert got[impl] == out[0][impl]
    assert got['refused'] is True
