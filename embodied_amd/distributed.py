"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

Sharding follows the reference's own multi-replica layout (ppo/main.py:190-192,
embodied/jax/internal.py:145-152): rank r owns the env block
[r*N, (r+1)*N) and the Replay those envs feed; payload never moves on insert.
The exchange steps are the two north_star names:

  * `all_gather_batch`  — RCCL all-gather of the sampled trajectories.  The
    local batch is sampled straight into ONE packed byte buffer (all keys, 256-B
    aligned), so the exchange is a single collective of B*L*S bytes per rank
    instead of one per key.
  * `all_reduce_mean`   — all-reduce of one flat gradient buffer
    (embodied/jax/opt.py:52-54's pmean).
  * `exchange_dp_slices` — SURVEY.md 8e's cheaper form of the trajectory
    exchange: every rank ends up with ITS slice of the global batch only, one
    all-to-all of B*L*S bytes per rank instead of an all-gather that hands every
    rank all (n-1) other batches.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

ALIGN = 256


def init(backend=None, device=None):
  """Join the job described by RANK / WORLD_SIZE / MASTER_* (torchrun)."""
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  if world > 1 and not dist.is_initialized():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    kwargs = {}
    if backend == 'nccl' and device is not None:
      kwargs['device_id'] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
  return rank, world


def world():
  return dist.get_world_size() if dist.is_initialized() else 1


def rank():
  return dist.get_rank() if dist.is_initialized() else 0


def env_block(n_global, r=None, w=None):
  """Global env ids owned by rank r: one contiguous block per rank."""
  r = rank() if r is None else r
  w = world() if w is None else w
  assert n_global % w == 0, (n_global, w)
  per = n_global // w
  return np.arange(r * per, (r + 1) * per, dtype=np.int64)


class PackedLayout:
  """Byte offsets of every key of a (B, L, ...) batch inside one flat buffer."""

  def __init__(self, specs, batch, length):
    # specs: [(name, torch dtype, per-step shape)]
    self.batch, self.length = batch, length
    self.entries = []
    offset = 0
    for name, dtype, shape in specs:
      itemsize = torch.empty((), dtype=dtype).element_size()
      nbytes = batch * length * itemsize * int(np.prod(shape, dtype=np.int64))
      self.entries.append((name, dtype, tuple(shape), offset, nbytes))
      offset += -(-nbytes // ALIGN) * ALIGN
    self.nbytes = offset
    self.index = {e[0]: e for e in self.entries}

  def view(self, flat, name):
    """Typed (B, L, ...) view of one key inside a 1-D packed buffer, or
    (G, B, L, ...) over a (G, nbytes) stack of packed buffers."""
    _, dtype, shape, offset, nbytes = self.index[name]
    if flat.dim() == 2:
      return flat[:, offset: offset + nbytes].view(dtype).view(
          flat.shape[0], self.batch, self.length, *shape)
    return flat[offset: offset + nbytes].view(dtype).view(self.batch, self.length, *shape)

  def views(self, flat, lead=None):
    """Typed views into `flat` ((nbytes,) or (world, nbytes) uint8)."""
    lead = (self.batch,) if lead is None else lead
    out = {}
    for name, dtype, shape, offset, nbytes in self.entries:
      if flat.dim() == 1:
        out[name] = flat[offset: offset + nbytes].view(dtype).view(
            *lead, self.length, *shape)
      else:
        part = flat[:, offset: offset + nbytes]
        out[name] = part.view(dtype).view(
            flat.shape[0], *lead, self.length, *shape)
    return out


class PackedViews(dict):
  """Per-key views of a packed batch, made on first access (a view costs ~3
  tensor ops on the host; a train step usually touches a few keys)."""

  def __init__(self, flat, layout):
    super().__init__()
    self._flat, self._layout = flat, layout

  def __missing__(self, name):
    value = self[name] = self._layout.view(self._flat, name)
    return value

  def keys(self):
    return self._layout.index.keys()

  def __iter__(self):
    return iter(self._layout.index)

  def __len__(self):
    return len(self._layout.index)

  def __contains__(self, name):
    return name in self._layout.index

  def items(self):
    return [(name, self[name]) for name in self._layout.index]

  def values(self):
    return [self[name] for name in self._layout.index]


class SampleInfo:
  """What `sample_packed` knows about one packed batch."""
  __slots__ = ('layout', 'online', 'nbytes', 'entries', 'batch', 'length')

  def __init__(self, layout, online):
    self.layout, self.online = layout, online
    self.nbytes, self.entries = layout.nbytes, layout.entries
    self.batch, self.length = layout.batch, layout.length

  def views(self, flat, lead=None):
    return self.layout.views(flat, lead)


def sample_packed(replay, batch, mode='train', groups=1, reuse=0):
  """`replay.sample` into one packed buffer: returns (flat uint8, per-key views,
  info with `.online` flags and the layout).  The gather kernel writes each key
  at its offset directly.

  `reuse` = K >= 2 rotates K buffers of the replay's own instead of allocating
  one per call (`Replay(reuse_outputs=K)`'s contract: a packed batch is
  overwritten by the K-th sample after it, on the sampling stream -- a caller that
  sends it from another stream waits for that send first, as `exchange` +
  `wait` one train step later do).

  `groups` > 1 cuts the batch into that many equal blocks of `batch / groups`
  sequences, each block a complete packed sub-batch (all keys) of its own:
  block d is what destination rank d receives in `exchange_dp_slices`.  The
  views are then shaped (groups, batch / groups, L, ...)."""
  import ctypes as C
  from . import _lib
  from .core import limiters
  assert batch % groups == 0, (batch, groups)
  if not replay._nonempty:
    limiters.wait(lambda: len(replay._native), f'Replay buffer {replay.name} is empty')
    replay._nonempty = True
  with replay._lock:
    replay._flush()
    cache = replay.__dict__.setdefault('_packed_layouts', {})
    layout = cache.get((batch, groups))
    if layout is None:
      layout = cache[(batch, groups)] = PackedLayout(
          [(k.name, k.dtype, k.shape) for k in replay._keys], batch // groups, replay.length)
      layout.offsets = [layout.index[k.name][3] for k in replay._keys]
      layout.ptrs = (C.c_void_p * len(replay._keys))()
      layout.groups = groups
      layout.online_ptrs = [np.zeros(batch, np.uint8) for _ in range(4)]   # handed out in turn
      layout.turn = 0
      layout.ring, layout.ring_turn = [], -1        # reuse=K: (buffer, key addresses) sets used in turn
    if reuse:
      assert reuse >= 2, reuse
      ring = layout.ring
      slot = layout.ring_turn = (layout.ring_turn + 1) % reuse
      if slot >= len(ring):
        flat = torch.empty((groups * layout.nbytes,), dtype=torch.uint8, device=replay.device)
        ptrs = (C.c_void_p * len(replay._keys))(*[flat.data_ptr() + o for o in layout.offsets])
        ring.append((flat, ptrs))
      flat, ptrs = ring[slot]
    else:
      flat = _lib.empty((groups * layout.nbytes,), torch.uint8, replay.device)
      base, ptrs = flat.data_ptr(), layout.ptrs
      for i, offset in enumerate(layout.offsets):
        ptrs[i] = base + offset
    layout.turn = (layout.turn + 1) & 3
    online = layout.online_ptrs[layout.turn]
    if groups == 1:
      _lib.fast.emb_replay_sample(
          replay._h, batch, _lib.MODES[mode], ptrs, _lib.ptr(online), None,
          replay._stream())
      views = PackedViews(flat, layout)
    else:
      _lib.fast.emb_replay_sample_grouped(
          replay._h, batch, _lib.MODES[mode], ptrs, batch // groups, layout.nbytes,
          _lib.ptr(online), None, replay._stream())
      views = PackedViews(flat.view(groups, layout.nbytes), layout)
  # (a private copy: the layout's scratch buffers are shared by every caller
  # with this batch shape and are written again four samples later)
  return flat, views, SampleInfo(layout, online.view(np.bool_).copy())


def gae_packed(flat, info, value, hor=200, lam=0.8, out=None):
  """`scans.gae` on a packed batch (dense or grouped — sampled locally or
  delivered by `exchange_dp_slices`) without materialising views or dense copies:
  the scan kernel reads `reward`, `is_last`, `is_terminal` in place.  `value` is
  the critic's dense (B, T) float32 output.  Returns adv, tar (B, T-1); `out` =
  (adv, tar), two contiguous float32 (B, T-1) tensors of the caller's, is
  written instead of fresh ones (`scans.gae(out=)`)."""
  from . import _lib, scans
  layout = info.layout
  groups = getattr(layout, 'groups', 1)
  B, T = layout.batch * groups, layout.length
  assert value.shape == (B, T) and value.dtype == torch.float32 and value.is_contiguous()
  base = flat.data_ptr()
  index = layout.index
  # The kernel reads raw bytes at these offsets: float32 rewards, 1-byte flags.
  assert index['reward'][1] == torch.float32, index['reward'][1]
  for name in ('is_last', 'is_terminal'):
    assert index[name][1] in (torch.bool, torch.uint8), (name, index[name][1])
  if out is None:
    both = _lib.empty((2, B, T - 1), torch.float32, flat.device)
    adv, tar = both.unbind(0)
  else:
    adv, tar = out
    for t in (adv, tar):
      assert t.shape == (B, T - 1) and t.dtype == torch.float32 and t.is_contiguous() and t.device == flat.device
  _lib.fast.emb_scan_gae_grouped(
      base + index['reward'][3], value.data_ptr(), base + index['is_last'][3],
      base + index['is_terminal'][3], B, T, scans._round32(1 - 1 / hor), scans._round32(lam),
      adv.data_ptr(), tar.data_ptr(), layout.batch if groups > 1 else 0,
      layout.nbytes if groups > 1 else 0, _lib.raw_stream(flat.device))
  return adv, tar


def _default_pg():
  from torch.distributed import distributed_c10d
  return distributed_c10d._get_default_group()


def async_all_gather(out, tensor, group=None):
  """Async all-gather into one flat tensor; returns a Work handle.  Calls the
  process group's own method: no Python-side argument checking, the GIL is
  released for the whole call (what matters when issued from CommThread)."""
  pg = group or _default_pg()
  if dist.get_backend(group) == 'gloo' and tensor.is_cuda:
    # (gloo moves host memory only: the test transport, see async_all_to_all)
    parts = [torch.empty(tensor.shape, dtype=tensor.dtype) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, tensor.cpu(), group=group)
    out.copy_(torch.cat(parts).view(out.dtype).reshape(out.shape))
    return _Finished()
  try:
    return pg._allgather_base(out, tensor)
  except AttributeError:
    return dist.all_gather_into_tensor(out, tensor, async_op=True)


def async_all_reduce(tensor, group=None):
  """Async in-place sum over ranks; returns a Work handle."""
  pg = group or _default_pg()
  try:
    return pg.allreduce([tensor])
  except AttributeError:
    return dist.all_reduce(tensor, async_op=True)


def async_all_to_all(out, tensor, group=None):
  """Async equal-split all-to-all: block d of `tensor` goes to rank d, block s
  of `out` comes from rank s.  Returns a Work handle (`.wait()`)."""
  pg = group or _default_pg()
  if dist.get_backend(group) == 'gloo' and tensor.is_cuda:
    # gloo moves host memory only (test transport: the multi-rank control flow
    # on a box with fewer GPUs than ranks); RCCL takes the device buffers as is.
    host_in, host_out = tensor.cpu(), torch.empty(out.shape, dtype=out.dtype)
    pg.alltoall_base(host_out, host_in, [], []).wait()
    out.copy_(host_out)
    return _Finished()
  return pg.alltoall_base(out, tensor, [], [])


class _Finished:
  def wait(self):
    return True


def exchange_dp_slices(flat, info):
  """All-to-all of a grouped packed batch (`sample_packed(..., groups=world)`):
  rank r keeps block r of its own batch and receives block r of every other
  rank's.  Returns (work, received flat, per-key views shaped (world, B/world,
  L, ...)): this rank's slice of the global batch, mixed from every rank's
  envs.  Each rank sends and receives (n-1)/n * B*L*S bytes; the all-gather
  form receives (n-1) * B*L*S."""
  w = world()
  out = torch.empty_like(flat)
  if w == 1:
    out.copy_(flat)
    work = _Finished()
  else:
    work = async_all_to_all(out, flat)
  return work, out, PackedViews(out.view(info.layout.groups, info.layout.nbytes), info.layout)


class Done:
  """`CommThread.submit` without the thread: runs `fn` now; `.result()` returns
  what it returned."""
  __slots__ = ('_value',)

  def __init__(self, fn):
    self._value = fn()

  def result(self):
    return self._value


class CommThread:
  """Issues collectives from a helper thread so their host cost (tens of
  microseconds each inside c10d, with the GIL released) does not sit on the
  thread that steps the envs.  One thread, FIFO: every rank issues the same
  collectives in the same order.  `submit(fn)` returns a future whose result is
  whatever `fn` returned (an async Work handle).  Drain (`wait_all`) before
  issuing any collective from another thread."""

  def __init__(self, device=None):
    import concurrent.futures
    import queue
    import threading
    self._Future = concurrent.futures.Future
    self._queue = queue.SimpleQueue()
    self._device = device
    self._thread = threading.Thread(target=self._loop, name='emb_comm', daemon=True)
    self._thread.start()

  def _loop(self):
    if self._device is not None and torch.device(self._device).type == 'cuda':
      torch.cuda.set_device(self._device)     # current device is per thread
    while True:
      item = self._queue.get()
      if item is None:
        return
      fn, future = item
      try:
        future.set_result(fn())
      except BaseException as e:              # hand every failure to the waiter
        future.set_exception(e)

  def submit(self, fn):
    future = self._Future()
    self._queue.put((fn, future))
    return future

  def close(self):
    self._queue.put(None)
    self._thread.join(timeout=10)


def all_gather_packed(flat, layout):
  """One all-gather of the packed local batch; returns per-key views shaped
  (world, B, L, ...) over the gathered buffer (no regroup copy)."""
  w = world()
  if w == 1:
    return layout.views(flat[None])
  out = torch.empty(w * flat.numel(), dtype=torch.uint8, device=flat.device)
  dist.all_gather_into_tensor(out, flat)
  return layout.views(out.view(w, flat.numel()))


def all_gather_batch(batch):
  """Generic form for any dict of same-leading-dim tensors: packs, gathers
  once, returns dict of (world*B, ...) tensors."""
  w = world()
  if w == 1:
    return dict(batch)
  names = list(batch)
  device = batch[names[0]].device
  sizes, offset = [], 0
  for k in names:
    nbytes = batch[k].numel() * batch[k].element_size()
    sizes.append((offset, nbytes))
    offset += -(-nbytes // ALIGN) * ALIGN
  flat = torch.empty(offset, dtype=torch.uint8, device=device)
  for k, (off, nbytes) in zip(names, sizes):
    flat[off: off + nbytes].copy_(batch[k].contiguous().view(torch.uint8).reshape(-1))
  out = torch.empty(w * offset, dtype=torch.uint8, device=device)
  dist.all_gather_into_tensor(out, flat)
  out = out.view(w, offset)
  result = {}
  for k, (off, nbytes) in zip(names, sizes):
    v = batch[k]
    part = out[:, off: off + nbytes].contiguous().view(v.dtype)
    result[k] = part.view(w * v.shape[0], *v.shape[1:])
  return result


def all_reduce_mean(flat):
  """In-place mean over ranks of one flat buffer (the gradient pmean)."""
  w = world()
  if w > 1:
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(w)
  return flat


def max_over_ranks(value, device):
  if world() == 1:
    return float(value)
  t = torch.tensor([value], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


class ShardedReplay:
  """One logical replay over the envs of ALL ranks.

  Every rank advances the same host index for every global worker (same seed,
  same insert order by global env id), so item ids, eviction and sampled indices
  are identical on all ranks and identical to a single-process replay over all
  envs.  Payload stays where it was produced: rank r holds only the chunk slots
  of its own env block.  `sample` gathers the sequences this rank owns into
  their final positions of a zero-filled packed (B, L, S) buffer and one RCCL
  all-reduce(sum) — supports are disjoint, so the sum is a merge — leaves the
  full batch on every rank.
  """

  def __init__(self, length, capacity, envs_per_rank, rank=None, world=None,
               reduce=None, **kwargs):
    from .core.replay import Replay
    self.rank = globals()['rank']() if rank is None else rank
    self.world = globals()['world']() if world is None else world
    self.envs_per_rank = envs_per_rank
    self.n_global = envs_per_rank * self.world
    self.workers = np.arange(self.n_global, dtype=np.int64)
    self.local = slice(self.rank * envs_per_rank, (self.rank + 1) * envs_per_rank)
    self._reduce = reduce or self._all_reduce
    slots = kwargs.pop('slots', None)
    if slots is None:
      chunksize = kwargs.get('chunksize', 1024)
      per_owner = -(-(capacity + length) // (chunksize * self.world)) + 2 * envs_per_rank + 4
      slots = per_owner * self.world
    self.replay = Replay(
        length, capacity, owners=self.world, owner=self.rank,
        workers_per_owner=envs_per_rank, slots=slots, **kwargs)
    self.length = self.replay.length

  def __len__(self):
    return len(self.replay)

  @staticmethod
  def _all_reduce(flat):
    if world() > 1:
      dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat

  def add_batch(self, steps):
    """One vectorised step of THIS rank's env block ((envs_per_rank, ...) device
    tensors).  All ranks call it in lockstep."""
    import ctypes as C
    from . import _lib
    from ._lib import api
    rep = self.replay
    steps = {k: v for k, v in steps.items() if not k.startswith('log/')}
    with rep._lock:
      if rep._keys is None:
        rep._init_keys({k: v[0] for k, v in steps.items()})
      n = self.n_global
      rows = np.zeros(n, np.int32)
      sids = np.zeros((n, _lib.STEPID_BYTES), np.uint8)
      api.emb_replay_add_index(
          rep._handle, n, _lib.ptr(self.workers), _lib.ptr(rows), _lib.ptr(sids), None)
      mine = np.ascontiguousarray(rows[self.local])
      sid_dev = torch.from_numpy(np.ascontiguousarray(sids[self.local])).to(
          rep.device, non_blocking=True)
      ids = (C.c_int32 * len(rep._keys))(*range(len(rep._keys)))
      ptrs = (C.c_void_p * len(rep._keys))()
      keep = [sid_dev]
      for i, key in enumerate(rep._keys):
        if key.name == 'stepid':
          ptrs[i] = sid_dev.data_ptr()
          continue
        value = steps[key.name].to(rep.device, key.dtype).contiguous()
        assert value.shape == (self.envs_per_rank, *key.shape), (key.name, value.shape)
        keep.append(value)
        ptrs[i] = value.data_ptr()
      api.emb_replay_scatter_rows(
          rep._handle, _lib.ptr(mine), len(mine), len(rep._keys), ids, ptrs, rep._stream())

  def sample(self, batch, mode='train'):
    """The same (batch, length, ...) dict on every rank."""
    import ctypes as C
    from . import _lib
    from ._lib import api
    from .core import limiters
    rep = self.replay
    limiters.wait(lambda: len(rep._native), f'Replay buffer {rep.name} is empty')
    with rep._lock:
      rows = np.zeros((batch, rep.length), np.int32)
      owners = np.zeros(batch, np.int64)
      api.emb_replay_sample_index(
          rep._handle, batch, _lib.MODES[mode], _lib.ptr(rows), None, _lib.ptr(owners))
      rows[owners // self.envs_per_rank != self.rank] = -1
      layout = PackedLayout(
          [(k.name, k.dtype, k.shape) for k in rep._keys], batch, rep.length)
      flat = torch.zeros(layout.nbytes, dtype=torch.uint8, device=rep.device)
      views = layout.views(flat)
      ptrs = (C.c_void_p * len(rep._keys))(*[views[k.name].data_ptr() for k in rep._keys])
      api.emb_replay_gather_rows(
          rep._handle, _lib.ptr(rows), rows.size, rep.length, ptrs, rep._stream())
    self._reduce(flat)
    return views


class GlobalClock:
  """Wall-clock decisions every replica takes alike (clock.py:77-94: log / save
  / report in lockstep).  The reference asks an RPC server that runs replica
  0's clock behind two barriers; here the k-th call on every rank is ONE MAX
  all-reduce of two flags -- "somebody asked to skip" and "rank 0's clock is
  due" -- over the process group (host tensors with gloo, device tensors with
  RCCL).  Same rules as the server's `should` (clock.py:44-67): 0 = never,
  negative = always, rank 0's clock restarts when due even if the decision is
  then vetoed by a skip; the first call is skipped unless `first`
  (clock.py:81-82,87-89).  Without a process group (or with one rank) it is a
  LocalClock."""

  def __init__(self, every, first=False, group=None, device=None):
    from .utils import LocalClock
    self.every = float(every)
    self.group = group
    self.multihost = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if not self.multihost:
      self.clock = LocalClock(every, first)
      return
    if dist.get_backend(group) == 'nccl':
      device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
    else:
      device = torch.device('cpu')
    self.device = device
    self.rank = dist.get_rank(group)
    # every replica must have built the clock with the same period (clock.py:38-39)
    periods = torch.tensor([self.every, -self.every], dtype=torch.float64, device=device)
    dist.all_reduce(periods, op=dist.ReduceOp.MAX, group=group)
    assert periods[0].item() == -periods[1].item() == self.every, 'GlobalClock: periods differ'
    import time
    self.prev = time.time()
    self.skip_next = not first

  def __call__(self, step=None, skip=None):
    if not self.multihost:
      return self.clock(step, skip)
    import time
    if self.skip_next:
      self.skip_next = False
      skip = True
    due = False
    if self.rank == 0:
      now = time.time()
      if self.every < 0:
        due = True
      elif self.every > 0 and now >= self.prev + self.every:
        self.prev = now
        due = True
    flags = torch.tensor([1.0 if skip else 0.0, 1.0 if due else 0.0], device=self.device)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
    skipped, due = flags.tolist()
    return bool(due) and not bool(skipped)


def clock_setup(is_server, replica, replicas, port, addr):
  """`embodied.clock.setup` by name (clock.py:11-23: connect this replica to the
  clock server).  Here the replicas' clocks meet on a torch.distributed group
  (GlobalClock above), so "connecting" = making sure that group exists: a gloo
  group at tcp://addr:port if the host program has not initialised one.  One
  replica: nothing to do, as in the reference (`replicas <= 1`)."""
  if replicas <= 1 or (dist.is_available() and dist.is_initialized()):
    return
  host = (addr or '127.0.0.1').split(':')[0]
  dist.init_process_group('gloo', init_method=f'tcp://{host}:{int(port)}', rank=int(replica),
                          world_size=int(replicas))


def pmean(x, comm=None, group=None):
  """Mean over the data-parallel ranks of a (small) float32 tensor, e.g. a local
  mean (embodied/jax/utils.py:76-81: `x.mean()` then `jax.lax.pmean`).  `comm`:
  a NativeComm (RCCL through the C ABI); else the process group; one rank: x."""
  x = x.to(torch.float32).contiguous()
  if comm is not None:
    return comm.pmean(x.clone())
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
    return x
  out = x.clone()
  dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
  return out / dist.get_world_size(group)


def percentile_over_ranks(x, q, comm=None, group=None):
  """`jnp.percentile(all_gather(x), q)` (embodied/jax/utils.py:83-88): the q-th
  percentile (linear interpolation) of every rank's values together.  Every
  rank must pass the same number of values.  q may be a list."""
  flat = x.reshape(-1).to(torch.float32).contiguous()
  if comm is not None:
    flat = comm.all_gather_returns(flat)
  elif dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
    gathered = torch.empty(dist.get_world_size(group) * flat.numel(), dtype=torch.float32,
                           device=flat.device)
    dist.all_gather_into_tensor(gathered, flat, group=group)
    flat = gathered
  qs = torch.as_tensor(q, dtype=torch.float32, device=flat.device) / 100
  if flat.numel() > 1 << 24:                 # torch.quantile's size limit: sort instead
    ordered = flat.sort().values
    pos = qs * (flat.numel() - 1)
    lo = pos.floor().long()
    hi = (lo + 1).clamp(max=flat.numel() - 1)
    return ordered[lo] + (ordered[hi] - ordered[lo]) * (pos - lo)
  return torch.quantile(flat, qs, interpolation='linear')


class Normalize:
  """Running normaliser statistics kept alike on every data-parallel rank
  (embodied/jax/utils.py:16-88): `impl` 'none' | 'meanstd' | 'perc'.  The
  cross-rank parts are `pmean` (means of the local batch) and
  `percentile_over_ranks` (percentiles of all ranks' values together);
  everything else is local float32 arithmetic in the reference's order."""

  def __init__(self, impl, rate=0.01, limit=1e-8, perclo=5.0, perchi=95.0, debias=True,
               comm=None, group=None):
    if impl not in ('none', 'meanstd', 'perc'):
      raise NotImplementedError(impl)
    self.impl, self.rate, self.limit = impl, rate, limit
    self.perclo, self.perchi, self.debias = perclo, perchi, debias
    self.comm, self.group = comm, group
    self.state = {}

  def _var(self, name, like):
    if name not in self.state:
      self.state[name] = torch.zeros((), dtype=torch.float32, device=like.device)
    return self.state[name]

  def _update(self, name, value, like):
    var = self._var(name, like)
    self.state[name] = (1 - self.rate) * var + self.rate * value

  def __call__(self, x, update=True):
    if update:
      self.update(x)
    return self.stats(x)

  def update(self, x):
    x = x.detach().to(torch.float32)
    if self.impl == 'meanstd':
      means = pmean(torch.stack([x.mean(), x.square().mean()]), self.comm, self.group)
      self._update('mean', means[0], x)
      self._update('sqrs', means[1], x)
    elif self.impl == 'perc':
      lo, hi = percentile_over_ranks(x, [self.perclo, self.perchi], self.comm, self.group)
      self._update('lo', lo, x)
      self._update('hi', hi, x)
    if self.debias and self.impl != 'none':
      self._update('corr', 1.0, x)

  def stats(self, like=None):
    if self.impl == 'none':
      return 0.0, 1.0
    if like is None:        # the device of the running statistics; before the first update: the default
      like = next(iter(self.state.values()), None)
      if like is None:
        like = torch.zeros((), device='cuda' if torch.cuda.is_available() else 'cpu')
    corr = 1.0
    if self.debias:
      corr = 1.0 / torch.clamp(self._var('corr', like), min=self.rate)
    if self.impl == 'meanstd':
      mean = self._var('mean', like) * corr
      std = torch.sqrt(torch.relu(self._var('sqrs', like) * corr - mean ** 2))
      return mean, torch.clamp(std, min=self.limit)
    lo, hi = self._var('lo', like) * corr, self._var('hi', like) * corr
    return lo, torch.clamp(hi - lo, min=self.limit)


class GroupComm:
  """NativeComm's train-step contract on a torch.distributed process group (RCCL,
  or gloo on a box with fewer GPUs than ranks): `exchange` issues the DP-slice
  all-to-all and / or the gradient all-reduce asynchronously, `wait` -- one
  train step later -- completes them.  Between the two calls the buffers belong
  to the exchange.  Same order of collectives on every rank: all-to-all first,
  then the all-reduce."""

  def __init__(self, group=None):
    self.group = group
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self._pending = []
    self._scale = None

  def exchange(self, slices=None, received=None, grads=None, mean=True, gather=False):
    assert not self._pending, 'GroupComm.exchange: wait() for the previous exchange first'
    if slices is not None and gather:
      # the trajectory all-gather: this rank's block -> world blocks in rank order
      assert received is not None and received.numel() == self.world * slices.numel()
      if self.world == 1:
        received.copy_(slices)
      else:
        self._pending.append(async_all_gather(received, slices, self.group))
    elif slices is not None:
      assert received is not None and received.numel() == slices.numel()
      assert slices.numel() % self.world == 0
      if self.world == 1:
        received.copy_(slices)
      else:
        self._pending.append(async_all_to_all(received, slices, self.group))
    if grads is not None and self.world > 1:
      if mean and dist.get_backend(self.group) == 'nccl':
        # RCCL averages in the collective itself (no second pass over the buffer)
        opts = dist.AllreduceOptions()
        opts.reduceOp = dist.ReduceOp.AVG
        self._pending.append((self.group or _default_pg()).allreduce([grads], opts))
      else:
        self._pending.append(async_all_reduce(grads, self.group))      # sum; the mean is taken in wait()
        self._scale = grads if mean else None

  def wait(self, device=None):
    for work in self._pending:
      work.wait()
    self._pending.clear()
    if self._scale is not None:
      self._scale.div_(self.world)
      self._scale = None

  def close(self):
    self.wait()


class NativeComm:
  """The two collectives on RCCL through the library's own C ABI
  (`emb_comm_*`, include/embodied_hip.h) instead of torch.distributed: for hosts
  that have no process group.  The 128-byte id made by rank 0 reaches the other
  ranks through `share(id_bytes) -> id_bytes` (default: a torch.distributed
  object broadcast if a group exists; world 1 needs none), or is given as `ident`.  Collectives are
  asynchronous on the caller's current stream."""

  def __init__(self, rank=0, world=1, device=None, share=None, ident=None):
    import ctypes as C
    from . import _lib
    from ._lib import api
    self._lib, self._api, self._C = _lib, api, C
    self.rank, self.world = int(rank), int(world)
    self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    torch.cuda.set_device(self.device)
    if ident is None:
      ident = self.unique_id() if self.rank == 0 else bytes(128)
      if self.world > 1:
        if share is None:
          def share(data):
            box = [data]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        ident = share(ident)
    # (`ident`: rank 0's id, already handed round by the caller -- a caller that must
    # not have a process-group call inside this constructor, bench.py's self-check)
    self._handle = C.c_void_p()
    api.emb_comm_init((C.c_uint8 * 128).from_buffer_copy(ident), self.rank, self.world, C.byref(self._handle))

  @staticmethod
  def unique_id():
    """A fresh 128-byte communicator id (rank 0 makes it, every rank passes the same
    one to the constructor).  Also loads RCCL: raises if it cannot be bound."""
    import ctypes as C
    from ._lib import api
    ident = (C.c_uint8 * 128)()
    api.emb_comm_unique_id(ident)
    return bytes(ident)

  def all_gather(self, flat, out=None):
    """(nbytes,) uint8 per rank -> (world * nbytes,) uint8 on every rank."""
    assert flat.dtype == torch.uint8 and flat.is_contiguous() and flat.is_cuda
    if out is None:
      out = torch.empty(self.world * flat.numel(), dtype=torch.uint8, device=flat.device)
    self._api.emb_comm_allgather_traj(
        self._handle, flat.data_ptr(), out.data_ptr(), flat.numel(),
        self._lib.raw_stream(flat.device))
    return out

  def all_reduce(self, grads, mean=True):
    """In-place sum or mean over ranks of a flat f16 / bf16 / f32 / f64 buffer."""
    code = {torch.float16: self._lib.F16, torch.bfloat16: self._lib.BF16,
            torch.float32: self._lib.F32, torch.float64: self._lib.F64}.get(grads.dtype)
    assert code is not None and grads.is_contiguous() and grads.is_cuda, grads.dtype
    self._api.emb_comm_allreduce_grads_as(
        self._handle, grads.data_ptr(), grads.numel(), code, int(bool(mean)),
        self._lib.raw_stream(grads.device))
    return grads

  def all_to_all(self, flat, out=None):
    """(world * nbytes,) uint8: block r goes to rank r; block r of the result
    came from rank r (the DP-slice exchange of `exchange_dp_slices`)."""
    assert flat.dtype == torch.uint8 and flat.is_contiguous() and flat.is_cuda
    assert flat.numel() % self.world == 0, (flat.numel(), self.world)
    if out is None:
      out = torch.empty_like(flat)
    self._api.emb_comm_alltoall_slices(
        self._handle, flat.data_ptr(), out.data_ptr(), flat.numel() // self.world,
        self._lib.raw_stream(flat.device))
    return out

  def all_gather_returns(self, values, out=None):
    """Every rank's float32 values, concatenated in rank order, on every rank
    (the `perc` normaliser's all-gather, embodied/jax/utils.py:83-88)."""
    assert values.dtype == torch.float32 and values.is_contiguous() and values.is_cuda
    if out is None:
      out = torch.empty(self.world * values.numel(), dtype=torch.float32, device=values.device)
    self._api.emb_comm_allgather_returns(
        self._handle, values.data_ptr(), out.data_ptr(), values.numel(),
        self._lib.raw_stream(values.device))
    return out

  def pmean(self, values):
    """In-place mean over the ranks of a small float32 tensor (Normalize._mean's
    pmean, embodied/jax/utils.py:76-81)."""
    assert values.dtype == torch.float32 and values.is_contiguous() and values.is_cuda
    self._api.emb_comm_pmean_scalars(
        self._handle, values.data_ptr(), values.numel(), self._lib.raw_stream(values.device))
    return values

  def exchange(self, slices=None, received=None, grads=None, mean=True, gather=False):
    """One train step's collectives on the communicator's own stream, after
    what the current stream holds so far: the DP-slice all-to-all of `slices`
    into `received` (both (world * nbytes,) uint8) -- or, with `gather`, the
    trajectory all-gather of `slices` (nbytes,) into `received` (world * nbytes,)
    -- and / or the in-place all-reduce of `grads`.  `wait()` orders the current
    stream after them; the caller keeps the buffers until then."""
    device = (grads if grads is not None else slices).device
    code, count = self._lib.F32, 0
    if grads is not None:
      code = {torch.float16: self._lib.F16, torch.bfloat16: self._lib.BF16,
              torch.float32: self._lib.F32, torch.float64: self._lib.F64}[grads.dtype]
      count = grads.numel()
    per_rank = 0
    if slices is not None and gather:
      assert slices.dtype == torch.uint8 and received is not None
      assert received.numel() == self.world * slices.numel()
      per_rank = slices.numel()
    elif slices is not None:
      assert slices.dtype == torch.uint8 and slices.numel() % self.world == 0
      assert received is not None and received.numel() == slices.numel()
      per_rank = slices.numel() // self.world
    (self._lib.fast.emb_comm_exchange_gather if gather else self._lib.fast.emb_comm_exchange)(
        self._handle.value, self._lib.raw_stream(device),
        slices.data_ptr() if per_rank else None, received.data_ptr() if per_rank else None, per_rank,
        grads.data_ptr() if count else None, count, code, int(bool(mean)))

  def wait(self, device=None):
    self._lib.fast.emb_comm_wait(self._handle.value, self._lib.raw_stream(device or self.device))

  def close(self):
    handle, self._handle = self._handle, None
    if handle:
      self._api.emb_comm_destroy(handle)

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


class DirectComm:
  """NativeComm's train-step contract on the library's DIRECT xGMI schedule
  (`emb_direct_*`, csrc/direct_comm.hip) instead of RCCL: every rank writes its
  peers' shares of the gradient all-reduce (reduce-scatter + all-gather) and of
  the DP-slice all-to-all straight into their memory through hipIpc handles, all
  n-1 peers at once -- one xGMI link each, where a ring is bound by one.  One
  node, at most 8 ranks.

  The 64-byte handle of every rank's buffer reaches the others through
  `share_all(handle_bytes) -> [handle_bytes of rank 0, 1, ...]` (default: an
  object all-gather on the torch.distributed group).  `max_grad_bytes` /
  `max_slice_bytes`: the largest gradient buffer and all-to-all / all-gather
  block (bytes per rank) this communicator will see.

  `timeout_ms` is a watchdog, not a latency bound: a kernel that waits longer
  than that for a peer gives up, writes NO result and kills the communicator --
  the next `exchange` / `wait` / collective on any rank raises (RCCL would keep
  waiting; a wrong gradient must never pass for a reduced one).  Default: ten
  minutes, as torch.distributed's own collective watchdog."""

  def __init__(self, rank=0, world=1, device=None, max_grad_bytes=64 << 20, max_slice_bytes=32 << 20,
               timeout_ms=600_000, share_all=None):
    import ctypes as C
    from . import _lib
    from ._lib import api
    self._lib, self._api, self._C = _lib, api, C
    self.rank, self.world = int(rank), int(world)
    self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    torch.cuda.set_device(self.device)
    self._handle = C.c_void_p()
    if self.world == 1:
      api.emb_direct_create(0, 1, int(max_grad_bytes), int(max_slice_bytes), int(timeout_ms), C.byref(self._handle))
      return
    if share_all is None:
      def share_all(data):
        box = [None] * self.world
        dist.all_gather_object(box, data)
        return box
    # A rank whose buffer cannot be made or mapped (no IPC between two of the
    # GPUs, out of memory) must not leave the others waiting inside share_all:
    # every rank takes part in both rounds whatever happened to it, and all of
    # them raise together.
    failure, mine = None, b''
    try:
      api.emb_direct_create(self.rank, self.world, int(max_grad_bytes), int(max_slice_bytes), int(timeout_ms),
                            C.byref(self._handle))
      raw = (C.c_uint8 * 64)()
      api.emb_direct_handle(self._handle, raw)
      mine = bytes(raw)
    except Exception as e:
      failure = e
    handles = share_all(mine)
    assert len(handles) == self.world, (len(handles), self.world)
    if failure is None and all(len(h) == 64 for h in handles):
      try:
        api.emb_direct_connect(self._handle, (C.c_uint8 * (64 * self.world)).from_buffer_copy(b''.join(handles)))
      except Exception as e:
        failure = e
    mapped = share_all(b'' if failure is not None else b'ok')
    if failure is not None or any(m != b'ok' for m in mapped):
      lost = [r for r, m in enumerate(mapped) if m != b'ok']
      self.close()
      raise RuntimeError(f'DirectComm: rank(s) {lost} could not set up the direct transport'
                         + (f' (here: {type(failure).__name__}: {failure})' if failure is not None else ''))

  _CODES = None

  def _code(self, dtype):
    codes = {torch.float16: self._lib.F16, torch.bfloat16: self._lib.BF16, torch.float32: self._lib.F32}
    assert dtype in codes, f'DirectComm: {dtype} gradients (f16, bf16, f32 only)'
    return codes[dtype]

  def all_reduce(self, grads, mean=True):
    """In-place sum or mean over the ranks of a flat f16 / bf16 / f32 buffer;
    every rank ends with the same bits.  The buffer starts on a 16-byte boundary
    (a view such as `flat[1:]` does not: reduce the whole allocation)."""
    assert grads.is_contiguous() and grads.is_cuda
    assert grads.data_ptr() % 16 == 0, 'DirectComm.all_reduce: the buffer must be 16-byte aligned'
    self._api.emb_direct_allreduce(
        self._handle, grads.data_ptr(), grads.numel(), self._code(grads.dtype), int(bool(mean)),
        self._lib.raw_stream(grads.device))
    return grads

  def all_to_all(self, flat, out=None):
    """(world * nbytes,) uint8: block r goes to rank r; block r of the result came from rank r."""
    assert flat.dtype == torch.uint8 and flat.is_contiguous() and flat.is_cuda
    assert flat.numel() % self.world == 0, (flat.numel(), self.world)
    if out is None:
      out = torch.empty_like(flat)
    self._api.emb_direct_alltoall(
        self._handle, flat.data_ptr(), out.data_ptr(), flat.numel() // self.world,
        self._lib.raw_stream(flat.device))
    return out

  def all_gather(self, flat, out=None):
    """(nbytes,) uint8 per rank -> (world * nbytes,) uint8 on every rank, rank order
    (the trajectory all-gather, embodied/jax/internal.py:145-152): this rank's
    block goes to all n-1 peers at once, one link each."""
    assert flat.dtype == torch.uint8 and flat.is_contiguous() and flat.is_cuda
    if out is None:
      out = torch.empty(self.world * flat.numel(), dtype=torch.uint8, device=flat.device)
    assert out.numel() == self.world * flat.numel()
    self._api.emb_direct_allgather(
        self._handle, flat.data_ptr(), out.data_ptr(), flat.numel(), self._lib.raw_stream(flat.device))
    return out

  def exchange(self, slices=None, received=None, grads=None, mean=True, gather=False):
    """One train step's collectives on the transport's own stream, after what
    the current stream holds so far (NativeComm.exchange's contract, `gather`
    included); `wait()` orders the current stream after them.  Raises if a wait
    inside an earlier operation timed out (the communicator is dead then)."""
    device = (grads if grads is not None else slices).device
    code, count = self._lib.F32, 0
    if grads is not None:
      code, count = self._code(grads.dtype), grads.numel()
      assert grads.data_ptr() % 16 == 0, 'DirectComm.exchange: the gradient buffer must be 16-byte aligned'
    per_rank = 0
    if slices is not None and gather:
      assert slices.dtype == torch.uint8 and received is not None
      assert received.numel() == self.world * slices.numel()
      per_rank = slices.numel()
    elif slices is not None:
      assert slices.dtype == torch.uint8 and slices.numel() % self.world == 0
      assert received is not None and received.numel() == slices.numel()
      per_rank = slices.numel() // self.world
    (self._lib.fast.emb_direct_exchange_gather if gather else self._lib.fast.emb_direct_exchange)(
        self._handle.value, self._lib.raw_stream(device),
        slices.data_ptr() if per_rank else None, received.data_ptr() if per_rank else None, per_rank,
        grads.data_ptr() if count else None, count, code, int(bool(mean)))

  def wait(self, device=None):
    self._lib.fast.emb_direct_wait(self._handle.value, self._lib.raw_stream(device or self.device))

  def set_timeout(self, timeout_ms):
    """The watchdog of the operations issued from now on."""
    self._api.emb_direct_set_timeout(self._handle, int(timeout_ms))

  def timed_out(self):
    """True if a wait inside one of the kernels gave up on a peer (synchronises)."""
    word = self._C.c_int32(0)
    self._api.emb_direct_status(self._handle, self._C.byref(word))
    return bool(word.value)

  def check(self):
    """Synchronise and raise if the transport is dead (for the end of a run, or
    before results of the last exchange are trusted without another call)."""
    if self.timed_out():
      raise RuntimeError('DirectComm: a wait for a peer timed out; the results since are invalid')

  def close(self):
    handle, self._handle = self._handle, None
    if handle:
      self._api.emb_direct_destroy(handle)

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
