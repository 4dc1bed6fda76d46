"""Replay buffer with the reference's interface (embodied/core/replay.py:14-127)
on an HBM-resident chunk pool.

Layout: for every key one device buffer of `n_slots * chunksize` rows of
`rowbytes` bytes — the reference's `Chunk` dict of numpy arrays
(embodied/core/chunk.py:43-47) flattened into one pool per key.  A step lives at
row `slot * chunksize + index`.  All integer state (items, FIFO, refcounts,
online queue, selector, PRNG) is in the host index core of libembodied_hip.so;
payload moves with three kernels: scatter (add), gather (sample, with the
is_first / is_last annotation fused) and scatter (update).

`sample()` returns torch tensors on the replay's device (set `numpy=True` for
host arrays like the reference's).  There is no CPU implementation.
"""
import concurrent.futures
import ctypes as C
import io
import os
import pathlib
import sys
import threading
import time
import weakref

import numpy as np
import torch

from .. import _lib
from .._lib import api, fast
from . import limiters
from . import selectors as selectorlib
from . import selectors     # `embodied.replay.selectors`, as ppo/main.py:202 spells it

# csrc/fastcall.c: Replay.add of a host step dict as one C call (None: Python path).
_add_step = getattr(fast.module, 'add_step', None)

_TORCH_OF = {
    np.dtype(np.uint8): torch.uint8, np.dtype(np.int8): torch.int8,
    np.dtype(np.int16): torch.int16, np.dtype(np.int32): torch.int32,
    np.dtype(np.int64): torch.int64, np.dtype(np.float16): torch.float16,
    np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
    np.dtype(np.bool_): torch.bool, np.dtype(np.uint16): torch.uint16,
    np.dtype(np.uint32): torch.uint32, np.dtype(np.uint64): torch.uint64,
}
_ITEMSIZE = {torch.bfloat16: 2}
_DTYPE_CODE = {
    torch.uint8: _lib.U8, torch.int8: _lib.I8, torch.int16: _lib.I16, torch.int32: _lib.I32,
    torch.int64: _lib.I64, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16,
    torch.float32: _lib.F32, torch.float64: _lib.F64, torch.bool: _lib.BOOL,
}


class Batch(dict):
  """A sampled batch: a dict name -> (batch, length, ...) tensor that remembers
  the addresses, shape and stream it was gathered with, so that
  `Replay.recycle(batch)` / `Replay.sample(..., out=batch)` cost no per-key
  checks.  Behaves like the plain dict the reference returns."""
  __slots__ = ('_emb_ptrs', '_emb_shape', '_emb_stream', '_emb_owner')


class _StepRecord:
  """What one vectorised step of a vector env looks like to the early insert and
  the publish behind it, remembered on the step's frame tensor.  Vector envs
  hand out the same few tensor objects step after step (an output ring): when
  every observation tensor IS the object the record was made for, and the agent
  stacks it the same way into the same staging tensor, all the per-key checks
  and pointer look-ups of the general path have been done before."""
  __slots__ = ('owner', 'values', 'workers', 'workers_ptr', 'n', 'frame_key', 'frames_ptr',
               'early_ptrs', 'memo', 'spec', 'out_ptr', 'publishes')


def _itemsize(dtype):
  return _ITEMSIZE.get(dtype) or torch.empty((), dtype=dtype).element_size()


class _Key:
  __slots__ = ('name', 'dtype', 'shape', 'rowbytes', 'pool', 'stage', 'stage_np', 'stages')

  def __init__(self, name, dtype, shape):
    self.name = name
    self.dtype = dtype              # torch dtype
    self.shape = tuple(shape)
    self.rowbytes = _itemsize(dtype) * int(np.prod(shape, dtype=np.int64))
    self.pool = None                # uint8 (rows * rowbytes,) on device
    self.stage = None               # pinned uint8 (stage_rows, rowbytes)


class Replay:

  def __init__(
      self, length, capacity=None, directory=None, chunksize=1024,
      online=False, selector=None, save_wait=False, name='unnamed', seed=0,
      device='cuda', numpy=False, slots=None, stage_rows=256, replica=0,
      owners=1, owner=0, workers_per_owner=0, reuse_outputs=0, heads=None):
    self.length = int(length)
    # heads = {name or 'prefix/': K}: `sample` returns only the first K steps of
    # those keys, (batch, K, ...) -- for keys whose consumer reads nothing else
    # (DreamerV3's replay context: dreamerv3/agent.py:322-331 takes x[:, :K] of the
    # sampled enc/ dyn/ dec/ entries, K = replay_context).  Every returned key
    # equals the full sample's [:, :K]; `update` still writes all T steps back.
    self._heads = {str(k): int(v) for k, v in (heads or {}).items()}
    for name, k in self._heads.items():
      if not 1 <= k <= self.length:
        raise ValueError(f'Replay(heads=): {name!r}: {k} is not in 1..length ({self.length})')
    self._key_lens = None           # ctypes int32[n_keys] once the keys are known (None: no key is cut)
    self.capacity = capacity and int(capacity)
    self.chunksize = int(chunksize)
    self.name = name
    self.online = online
    self.device = torch.device(device)
    if self.device.type == 'cuda' and self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    if self.device.type != 'cuda':
      raise RuntimeError(
          'embodied_amd.Replay keeps its chunk pool in HBM and moves it with '
          f'HIP kernels; device={device!r} is not a GPU (no CPU fallback).')
    self.numpy = numpy
    self.save_wait = save_wait
    directory = self.directory = _as_path(directory)
    # `selector or Uniform(seed)` in the reference (replay.py:26) silently drops
    # an empty selector that defines __len__; test for None instead.
    if selector is None:
      selector = selectorlib.Uniform(seed)
    self.sampler = selector
    self._native = selectorlib._as_native(selector)
    self._foreign_selector = isinstance(self._native, selectorlib.Foreign)   # Python callbacks: errors to re-raise
    if slots is None:
      slots = 64
      if self.capacity:
        slots = -(-(self.capacity + self.length) // self.chunksize) + 130
    # Sharded pools (distributed.ShardedReplay): the index covers every owner's
    # workers, this process holds only owner `owner`'s slot range in HBM.
    self._owners, self._owner = int(owners), int(owner)
    if self._owners > 1:
      slots = -(-int(slots) // self._owners) * self._owners
    self._slots = int(slots)
    cfg = _lib.ReplayConfig(
        self.length, self.capacity or 0, self.chunksize, self._slots,
        int(bool(online)), 0, int(replica), self._owners, int(workers_per_owner))
    self._handle = C.c_void_p()
    api.emb_replay_create(
        C.byref(cfg), self._native._handle, int(seed), C.byref(self._handle))
    self._h = self._handle.value      # plain int for the call shim
    self._keys = None
    self._keyid = {}
    self._lock = threading.RLock()
    self._stage_rows = int(stage_rows)
    self._staged = 0
    self._stage_dst = np.zeros(self._stage_rows, np.int32)
    self._one_worker = np.zeros(1, np.int64)
    self._one_row = np.zeros(1, np.int32)
    self._one_sid = np.zeros((1, _lib.STEPID_BYTES), np.uint8)
    # ndarray.ctypes.data costs ~1.2 us per access: addresses of the persistent
    # buffers are taken once.
    self._one_ptrs = (_lib.ptr(self._one_worker), _lib.ptr(self._one_row), _lib.ptr(self._one_sid))
    self._stage_plan = None
    self._many_row = np.zeros(self._stage_rows, np.int32)
    self._many_sid = np.zeros((self._stage_rows, _lib.STEPID_BYTES), np.uint8)
    self._many_ptrs = (self._many_row.ctypes.data, self._many_sid.ctypes.data)
    self._new_chunks = C.c_int32()
    self._new_chunks_ref = C.byref(self._new_chunks)
    self._pending_count = C.c_int64()
    self._pending_ref = C.byref(self._pending_count)
    self._saved = set()
    self._foreign = {}              # chunk ids of files other writers left -> local serials
    self._updates = 0
    self._workers_seen = None
    self._last_stream = None
    self._multistream = False
    self._workers_np = None
    self._replica = int(replica)
    # reuse_outputs=K: `sample` hands out K rotating sets of output tensors per
    # batch shape instead of fresh ones (a batch stays valid for K-1 further
    # samples): no allocations, and the gather writes into cache-warm lines.
    self._reuse = int(reuse_outputs)
    self._out_ring = {}
    self._templates = {}
    self._mask_plans = {}
    self._add_plan = None
    self._stage_busy, self._stage_pending = None, False
    self._stage_events, self._stage_set = [None, None], 0
    self._rowbytes_total = None
    # Output sets handed back with `recycle`.
    self._free = {}
    self._nonempty = False
    self._full = False
    self._len_out = C.c_int64()
    self._len_ref = C.byref(self._len_out)
    # Early insert (offer / _early_insert): the Driver's offer of the current
    # step, the token of the launch that took it up, cached plans.
    self._offer_tag = weakref.ref(self)
    self._offer_obs, self._offer_workers = None, None
    self._offer_names, self._early_plans, self._early_specs = {}, {}, {}
    self._colspecs = {}
    self._early_ptrs = None
    self._pre_token = 0
    self._cur_rec = None            # the step record the standing early insert went through
    self._carrying, self._carry_keep, self._carry_tmp = False, None, None
    self._token = C.c_uint64()
    self.early_inserts = 0          # steps whose observation keys went in with the obs stack
    self._savers = concurrent.futures.ThreadPoolExecutor(16, 'replay_saver')
    if directory and self._owners == 1 and pathlib.Path(directory).is_dir():
      self._reserve_directory_uids(directory)

  def __del__(self):
    handle, self._handle = getattr(self, '_handle', None), None
    if handle is not None and api is not None:
      api.raw.emb_replay_destroy(handle)

  # ------------------------------------------------------------------ state --

  def __len__(self):
    # A replay with a capacity that has filled up stays at its capacity: an
    # insert evicts first (replay.py:171-179) and nothing else removes items.
    # Run loops ask on every env step (run/train.py:70): no library call then
    # (which would also wait for the helper thread's bookkeeping of the step
    # just published).
    if self._full:
      return self.capacity
    n = self._len_out
    api.emb_replay_len(self._handle, self._len_ref)
    if self.capacity and n.value == self.capacity and self._owners == 1:
      self._full = True
    return n.value

  def online_pending(self):
    """Fresh on-policy windows queued for the next train-mode samples (online
    mode, replay.py:114-118)."""
    n = self._pending_count
    api.emb_replay_online_pending(self._handle, self._pending_ref)
    return n.value

  def _stream(self):
    stream = _lib.raw_stream(self.device)
    if stream != self._last_stream:
      # A second stream appeared (actor / learner split): from now on order
      # pool writes and reads across streams.
      if self._last_stream is not None and not self._multistream:
        api.emb_replay_multistream(self._handle, 1)
        self._multistream = True
      self._last_stream = stream
    return stream

  def stats(self):
    """replay.py:58-74 (counters reset on read)."""
    out = np.zeros(6, np.int64)
    api.emb_replay_stats(self._handle, _lib.ptr(out), 1)
    items, chunks, streams, inserts, samples, _ = out.tolist()
    updates, self._updates = self._updates, 0
    rowbytes = sum(k.rowbytes for k in self._keys) if self._keys else 0
    return {
        'items': items,
        'chunks': chunks,
        'streams': streams,
        'ram_gb': chunks * self.chunksize * rowbytes / 1024 ** 3,
        'inserts': inserts,
        'samples': samples,
        'updates': updates,
        'replay_ratio': self.length * samples / inserts if inserts else np.nan,
    }

  # ----------------------------------------------------------------- schema --

  def _init_keys(self, example):
    """Keys, dtypes and shapes are fixed by the first step (chunk.py:43-47)."""
    keys = []
    for name, value in example.items():
      if torch.is_tensor(value):
        keys.append(_Key(name, value.dtype, value.shape))
      else:
        value = np.asarray(value)
        if value.dtype not in _TORCH_OF:
          raise TypeError(f'replay key {name!r}: unsupported dtype {value.dtype}')
        keys.append(_Key(name, _TORCH_OF[value.dtype], value.shape))
    keys.append(_Key('stepid', torch.uint8, (_lib.STEPID_BYTES,)))
    rows = self._slots // self._owners * self.chunksize
    for key in keys:
      key.pool = torch.empty(rows * key.rowbytes, dtype=torch.uint8, device=self.device)
      # Two pinned sets used in turn: the adds after a flush fill the other set
      # while the flush's H2D copies still read this one.
      both = torch.empty((2, self._stage_rows, key.rowbytes), dtype=torch.uint8).pin_memory()
      key.stages = [(both[i], both[i].numpy()) for i in range(2)]
      key.stage, key.stage_np = key.stages[0]
    self._keys = keys
    self._keyid = {k.name: i for i, k in enumerate(keys)}
    lens = [self._head_of(k.name) for k in keys]
    if any(n != self.length for n in lens):
      self._key_lens = (C.c_int32 * len(keys))(*lens)
    self._batch_ptrs = (C.c_void_p * len(keys))()
    self._push_keys()
    self._stage_set = 0
    self._stage_plans = [self._make_stage_plan(i) for i in range(2)]
    self._stage_plan = self._stage_plans[0]

  def _head_of(self, name):
    """Steps of a sampled sequence that `sample` returns for key `name`."""
    if name in ('stepid', 'is_first', 'is_last'):
      return self.length            # bookkeeping keys (update, the annotation's consumers) stay whole
    best, steps = -1, self.length
    for pattern, k in self._heads.items():
      hit = name == pattern or (pattern.endswith('/') and name.startswith(pattern))
      if hit and len(pattern) > best:
        best, steps = len(pattern), k
    return steps

  def _make_stage_plan(self, which):
    """`add`'s per-step work as one C call (fastcall.c stage_plan / add_step):
    None without the call shim, for more than 64 keys or for a dtype numpy
    cannot hand over as it is (bfloat16)."""
    if _add_step is None or len(self._keys) > 65:
      return None
    entries = []
    for key in self._keys[:-1]:
      try:
        dtype = np.dtype(_numpy_of(key.dtype))
      except Exception:
        return None
      kind = {'b': 'b', 'i': 'i', 'u': 'u', 'f': 'f'}.get(dtype.kind)
      if kind is None or key.dtype == torch.bfloat16:
        return None
      entries.append((sys.intern(key.name), key.stages[which][0].data_ptr(), key.rowbytes, kind,
                      dtype.itemsize, tuple(int(d) for d in key.shape)))
    sid = self._keys[-1]
    self._add_step_args = (
        C.cast(_lib.lib.emb_replay_add_index, C.c_void_p).value, self._h,
        *self._one_ptrs, C.addressof(self._new_chunks))
    return _lib.fast.module.stage_plan(
        tuple(entries), sid.stages[which][0].data_ptr(), sid.rowbytes, self._stage_dst.ctypes.data)

  def _push_keys(self):
    n = len(self._keys)
    names = (C.c_char_p * n)(*[k.name.encode() for k in self._keys])
    rowbytes = (C.c_int64 * n)(*[k.rowbytes for k in self._keys])
    # With sharded pools the library sees a virtual base: only this owner's row
    # range [first, first + rows) is ever dereferenced.
    first = self._owner * (self._slots // self._owners) * self.chunksize
    pools = (C.c_void_p * n)(*[
        k.pool.data_ptr() - first * k.rowbytes for k in self._keys])
    api.emb_replay_set_keys(self._handle, n, names, rowbytes, pools)

  def _grow(self, at_least=0):
    """Pool exhausted: double it, copying the live rows device-to-device."""
    if self._owners > 1:
      raise _lib.PoolFull(_lib.ERR_POOL_FULL, 'a sharded replay pool cannot grow: raise `slots`')
    new_slots = max(2 * self._slots, self._slots + at_least + 2)
    if self._carrying:
      api.emb_replay_settle(self._handle)      # a carried publish targets the OLD pool: write it before the copy
    if self._keys is not None:
      # The copy below runs on the caller's stream and the old pool goes back to
      # the allocator: with actor and learner on different streams, write-backs
      # still queued on the other stream would miss the copy and a gather could
      # read freed memory.  Growth is rare: drain the device first.
      if self._multistream:
        torch.cuda.synchronize(self.device)
      rows = new_slots * self.chunksize
      for key in self._keys:
        bigger = torch.empty(rows * key.rowbytes, dtype=torch.uint8, device=self.device)
        bigger[:key.pool.numel()].copy_(key.pool)
        key.pool = bigger
      pools = (C.c_void_p * len(self._keys))(*[k.pool.data_ptr() for k in self._keys])
      api.emb_replay_grow(self._handle, new_slots, pools)
      if self._multistream:
        torch.cuda.synchronize(self.device)     # the copies are done before anyone uses the new pool
    else:
      api.emb_replay_grow(self._handle, new_slots, None)
    self._slots = new_slots

  # -------------------------------------------------------------------- add --

  def add(self, step, worker=0):
    """One step of one worker stream (replay.py:77-118).  Host values are
    staged in pinned memory and written to HBM by one scatter launch per
    `stage_rows` steps (or at the next sample/update); device tensors go through
    `add_batch`."""
    plan = self._stage_plan
    if plan is not None:
      # The whole step in one C call (csrc/fastcall.c add_step): values checked
      # against the schema, index bookkeeping, copies into the pinned stage
      # rows.  -1: a value the C side does not take as it is (no buffer, another
      # dtype, a device tensor, a key or shape mismatch) -- nothing was touched,
      # the Python path below converts or raises.
      with self._lock:
        self._one_worker[0] = worker
        while True:
          if self._stage_pending:             # a flush still reads this set's pinned rows
            self._stage_busy.synchronize()
            self._stage_pending = False
          status = _add_step(self._stage_plan, step, self._staged, *self._add_step_args)
          if status != _lib.ERR_POOL_FULL:
            break
          self._flush()                       # (moves on to the other stage set)
          self._grow()
        if status == 0:
          self._reraise()
          if self._new_chunks.value and self._staged:
            self._flush_ahead_of(1)
          self._staged += 1
          if self._staged == self._stage_rows:
            self._flush()
          return
        if status > 0:
          _lib.check(status)
    step = {k: v for k, v in step.items() if not k.startswith('log/')}
    if any(torch.is_tensor(v) and v.is_cuda for v in step.values()):
      batch = {k: (v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v)))[None]
               for k, v in step.items()}
      return self.add_batch(batch, [worker])
    with self._lock:
      if self._keys is None:
        self._init_keys(step)
      # Validate and convert BEFORE touching the index: a rejected step must
      # leave no trace (no row without payload).
      if len(step) + 1 != len(self._keys):
        raise KeyError(f'replay step keys {sorted(step)} differ from the first step')
      rows = []
      for name, value in step.items():
        i = self._keyid.get(name)
        if i is None:
          raise KeyError(f'replay step key {name!r} was not in the first step')
        key = self._keys[i]
        value = np.asarray(value)
        if value.shape != key.shape:
          raise ValueError((name, value.shape, key.shape))
        rows.append((key, np.ascontiguousarray(value).astype(
            _numpy_of(key.dtype), copy=False).reshape(-1).view(np.uint8)))
      self._one_worker[0] = worker
      while True:
        try:
          api.emb_replay_add_index(
              self._handle, 1, *self._one_ptrs, self._new_chunks_ref)
          break
        except _lib.PoolFull:
          self._flush()
          self._grow()
      self._reraise()
      if self._new_chunks.value and self._staged:
        self._flush()                         # a recycled slot: what waited goes out ahead of this row
      slot = self._staged
      if self._stage_pending:                 # the last flush still reads the pinned rows
        self._stage_busy.synchronize()
        self._stage_pending = False
      for key, data in rows:
        key.stage_np[slot] = data
      self._keys[-1].stage_np[slot] = self._one_sid[0]
      self._stage_dst[slot] = self._one_row[0]
      self._staged += 1
      if self._staged == self._stage_rows:
        self._flush()

  def _flush_ahead_of(self, count):
    """Rows [staged, staged + count) have just been written into the stage by a
    call that opened a chunk in a RECYCLED slot: the `staged` rows before them
    may still hold rows of the chunk that had the slot, and one scatter must not
    carry two writers of a pool row.  Flush the earlier rows by themselves and
    carry the new ones over to the front of the other stage set."""
    first = self._staged
    old = [key.stage_np for key in self._keys]
    dst = self._stage_dst[first: first + count].copy()
    self._flush()                             # rows [0, first); moves on to the other set
    if self._stage_pending:
      self._stage_busy.synchronize()
      self._stage_pending = False
    for key, rows in zip(self._keys, old):
      key.stage_np[:count] = rows[first: first + count]
    self._stage_dst[:count] = dst

  def _flush(self):
    """Pinned staging -> HBM: one H2D copy per key, one scatter launch."""
    n = self._staged
    if not n:
      return
    srcs = []
    for key in self._keys:
      srcs.append(key.stage[:n].to(self.device, non_blocking=True))
    ids = (C.c_int32 * len(self._keys))(*range(len(self._keys)))
    ptrs = (C.c_void_p * len(self._keys))(*[s.data_ptr() for s in srcs])
    rows = self._stage_dst[:n].copy()
    api.emb_replay_scatter_rows(
        self._handle, _lib.ptr(rows), n, len(self._keys), ids, ptrs, self._stream())
    # The pinned rows of this set are reused after the NEXT flush: the first add
    # into a set waits for the H2D copies that last read it (an event, not a
    # stream drain -- the host goes on stepping envs), which by then are one
    # whole stage of adds old.
    turn = self._stage_set
    if self._stage_events[turn] is None:
      self._stage_events[turn] = torch.cuda.Event()
    self._stage_events[turn].record()
    turn = self._stage_set = 1 - turn
    for key in self._keys:
      key.stage, key.stage_np = key.stages[turn]
    self._stage_plan = self._stage_plans[turn]
    self._stage_busy = self._stage_events[turn]
    self._stage_pending = self._stage_busy is not None
    self._staged = 0

  def add_batch(self, steps, workers, mask=None):
    """N steps, one per listed worker, from (N, ...) arrays or device tensors:
    one host call, one scatter launch.  Equivalent to N `add` calls in order.

    `mask=(names, is_last)` fuses the Driver's action mask into the insert
    (driver.py:72-74): the listed keys are stored as `value * ~is_last` in their
    own dtype and the masked tensors are returned (dict name -> tensor);
    `mask=(names, is_last, outs)` writes them into the caller's `outs[name]`;
    `outs=False`: the masked values go to the pool rows only, None is returned."""
    workers, workers_ptr = self._workers_of(workers)
    n = len(workers)
    with self._lock:
      if self._keys is None:
        self._init_keys({
            k: (v[0] if torch.is_tensor(v) else np.asarray(v)[0])
            for k, v in steps.items() if not k.startswith('log/')})
      if (mask is None and self._stage_plan is not None and not self._pre_token
          and n <= self._stage_rows and _all_host_arrays(steps)):
        # Host arrays take the staging of `add` (n rows at once, one C call, the
        # payload leaves with the next flush) instead of one H2D copy per key.
        if self._staged + n > self._stage_rows:
          self._flush()
        fn, handle, _, _, _, new_chunks = self._add_step_args
        while True:
          if self._stage_pending:
            self._stage_busy.synchronize()
            self._stage_pending = False
          status = _add_step(self._stage_plan, steps, self._staged, fn, handle, workers_ptr,
                             *self._many_ptrs, new_chunks, n)
          if status != _lib.ERR_POOL_FULL:
            break
          self._flush()
          self._grow(2 * n)
        if status == 0:
          self._reraise()
          if self._new_chunks.value and self._staged:
            self._flush_ahead_of(n)
          self._staged += n
          if self._staged == self._stage_rows:
            self._flush()
          return None
        if status > 0:
          _lib.check(status)
      if self._staged:
        self._flush()
      keyid, keys, device = self._keyid, self._keys, self.device
      ptrs = self._batch_ptrs
      # Per (key order, n): which pool column each dict position feeds and what a
      # ready-to-use value looks like there (checked per call, looked up once).
      order = tuple(steps)
      plan = self._add_plan
      if plan is None or plan[0] != order or plan[1] != n:
        plan = self._add_plan = (order, n, self._plan_columns(order, n))
      keep = []
      todo = self._collect(steps, plan[2], ptrs)
      for value, (i, dtype, shape, name, *_) in todo or ():
        if not torch.is_tensor(value):
          value = torch.from_numpy(np.ascontiguousarray(value))
        if tuple(value.shape) != shape:
          raise ValueError((name, tuple(value.shape), shape))
        value = value.to(device, dtype, non_blocking=True).contiguous()
        keep.append(value)
        ptrs[i] = value.data_ptr()
      masked = None
      if mask is not None:
        names, flags, *given = mask
        plan = self._mask_plans.get(names)
        if plan is None:
          ids = (C.c_int32 * len(names))(*[keyid[name] for name in names])
          codes = (C.c_int32 * len(names))(*[_DTYPE_CODE[keys[keyid[name]].dtype] for name in names])
          plan = self._mask_plans[names] = (ids, codes, (C.c_void_p * len(names))())
        ids, codes, outs = plan
        if given and given[0] is False:
          # nobody wants the masked values back (a Driver whose env takes the
          # unmasked actions with `reset`): pool rows only.  The library may read
          # the sources at its NEXT launch (carried publish): converted copies
          # made above live until this replay's next such step.
          masked = None
          self._carry_tmp = keep
          for j in range(len(names)):
            outs[j] = None
        else:
          masked = given[0] if given and given[0] is not None else {}
          for j, name in enumerate(names):
            out = masked.get(name)
            if out is None:
              key = keys[ids[j]]
              out = masked[name] = _lib.empty((n, *key.shape), key.dtype, device)
            outs[j] = out.data_ptr()
        if flags.device != device or not flags.is_contiguous():
          flags = flags.to(device).contiguous()
        keep.append(flags)
      # The early insert of this step (offer / _early_insert), if there was one.
      token, self._pre_token, self._offer_obs = self._pre_token, 0, None
      self._cur_rec = None
      while True:
        try:
          if token:
            if mask is None:
              fast.emb_replay_publish(
                  self._h, n, workers_ptr, ptrs, 0, None, None, None, None, token, self._stream())
            else:
              fast.emb_replay_publish(
                  self._h, n, workers_ptr, ptrs, len(ids), ids, codes, outs, flags.data_ptr(),
                  token, self._stream())
          elif mask is None:
            fast.emb_replay_add(self._h, n, workers_ptr, ptrs, self._stream())
          else:
            fast.emb_replay_add_masked(
                self._h, n, workers_ptr, ptrs, len(ids), ids, codes, outs,
                flags.data_ptr(), self._stream())
          break
        except _lib.PoolFull:
          self._grow(2 * n)
      self._reraise()
    return masked

  def carry_publish(self, enable=True):
    """Let a publish whose only remaining key is a small masked one, and whose
    masked values nobody wants back (`add_step(..., masked=False)`), skip its
    launch: the key is written by this replay's NEXT launch -- the following
    step's early insert takes it along, anything else that touches the pool
    settles it first (emb_replay_carry_publish, include/embodied_hip.h).  One
    dependent launch less per env step.  The caller keeps the key's source
    tensor and the flags unchanged until then (the Driver does)."""
    api.emb_replay_carry_publish(self._handle, int(bool(enable)))
    self._carrying = bool(enable)

  def _workers_of(self, workers):
    """(private int64 copy, its address) of a worker list."""
    if workers is not self._workers_seen:
      self._workers_np = np.array(workers, np.int64, ndmin=1)      # private copy
      self._workers_ptr = _lib.ptr(self._workers_np)
      # Remember the caller's object only if it cannot change under us (the
      # Driver's read-only id array, a tuple): a list or writable array that is
      # mutated in place must be read again on every call.
      frozen = isinstance(workers, tuple) or (
          isinstance(workers, np.ndarray) and not workers.flags.writeable)
      self._workers_seen = workers if frozen else None
    return self._workers_np, self._workers_ptr

  # ----------------------------------------------------------- early insert --

  def offer(self, obs, workers):
    """Driver -> Replay before the policy runs (driver.py:65-69): `obs` holds
    the (N, ...) device tensors of the step that the next `add_batch(...,
    workers)` will bring back.  If the agent turns one of its uint8 frame keys
    into the policy batch with `ops.obs_stack` while the offer stands, that
    launch also writes the observation keys into the step's pool rows
    (emb_replay_obs_stack_insert: every frame is read once) and `add_batch` has
    only the actions left.  Without that call nothing changes."""
    self._offer_obs = obs
    self._offer_workers = workers
    self._pre_token = 0
    self._cur_rec = None
    order = tuple(obs)
    names = self._offer_names.get(order)
    if names is None:
      names = self._offer_names[order] = tuple(
          k for k, v in obs.items()
          if torch.is_tensor(v) and v.dtype is torch.uint8 and v.ndim == 4 and v.is_cuda)
    tag = self._offer_tag
    for name in names:
      frames = obs[name]
      if getattr(frames, '_emb_offer', None) is not tag:
        frames._emb_offer = tag

  def _early_insert(self, frames, pixels, channels, first, dtype, scale, offset, out, memo=None):
    """ops.obs_stack on an offered frame tensor.  True: the policy batch has
    been written (with or without the early insert); False: not handled.
    `memo`: ops.obs_stack's note on the frame tensor (same object = same options
    and staging tensor as when it was made)."""
    obs = self._offer_obs
    if obs is None or self._pre_token or self._keys is None:
      return False
    rec = frames.__dict__.get('_emb_rec')
    if (rec is not None and memo is not None and rec.memo is memo and rec.owner is self._offer_tag
        and self._offer_workers is rec.workers and fast.same_values(obs, rec.values)):
      # This step is, object for object, one that went through the general path
      # below before: straight to the launch.
      with self._lock:
        if not self._staged:
          fast.emb_replay_obs_stack_insert(
              self._h, rec.n, rec.workers_ptr, rec.frame_key, rec.frames_ptr, rec.spec, rec.out_ptr,
              rec.early_ptrs, self._stream(), self._token)
          self._pre_token = self._token.value
          if self._pre_token:
            self.early_inserts += 1
            self._cur_rec = rec
          return True
    with self._lock:
      if self._staged:
        self._flush()
      order = tuple(obs)
      frame_name = None
      for name in self._offer_names.get(order, ()):
        if obs[name] is frames:
          frame_name = name
          break
      if frame_name is None or frame_name not in self._keyid:
        return False
      n = frames.shape[0]
      plan = self._early_plans.get((order, frame_name, n))
      if plan is None:
        # Per position of the obs dict: (pool column or -1, dtype, (n, *shape),
        # name).  The frame key and the narrow keys (<= 256 bytes per step)
        # take part; everything else waits for add_batch.
        columns = []
        for name in order:
          i = self._keyid.get(name, -1)
          if i >= 0 and name != frame_name and self._keys[i].rowbytes > 256:
            i = -1
          columns.append(self._column(i, n, name))
        plan = self._early_plans[(order, frame_name, n)] = (
            tuple(columns), self._keyid[frame_name])
      columns, frame_key = plan
      if self._early_ptrs is None:
        self._early_ptrs = (C.c_void_p * len(self._keys))()
      if self._collect(obs, columns, self._early_ptrs):
        return False                     # some key is not a ready device tensor
      spec_key = (pixels, channels, first, dtype, scale, offset)
      spec = self._early_specs.get(spec_key)
      if spec is None:
        code = _DTYPE_CODE.get(dtype)
        if code not in (_lib.U8, _lib.F16, _lib.BF16, _lib.F32):
          return False
        struct = _lib.ObsSpec(
            pixels, channels, _lib.LAYOUT_CHANNELS_FIRST if first else _lib.LAYOUT_SAME,
            code, scale, offset)
        spec = self._early_specs[spec_key] = (struct, C.addressof(struct))
      _, workers_ptr = self._workers_of(self._offer_workers)
      if len(self._workers_np) != n:
        return False
      fast.emb_replay_obs_stack_insert(
          self._h, n, workers_ptr, frame_key, frames.data_ptr(), spec[1], out.data_ptr(),
          self._early_ptrs, self._stream(), self._token)
      self._pre_token = self._token.value
      if self._pre_token:
        self.early_inserts += 1
        if memo is not None and self._workers_seen is self._offer_workers:
          # Remember the step on its frame tensor (_StepRecord): the next time
          # these very tensors come round, the checks above are skipped.
          rec = _StepRecord()
          rec.owner, rec.values = self._offer_tag, tuple(obs.values())
          rec.workers, rec.workers_ptr, rec.n = self._offer_workers, workers_ptr, n
          rec.frame_key, rec.frames_ptr = frame_key, frames.data_ptr()
          rec.early_ptrs = (C.c_void_p * len(self._keys))(*self._early_ptrs)
          rec.memo, rec.spec, rec.out_ptr = memo, spec[1], out.data_ptr()
          rec.publishes = {}
          frames._emb_rec = rec
          self._cur_rec = rec
    return True

  def add_step(self, obs, acts, outs, workers, flags, masked):
    """The Driver's insert of one vectorised step whose observations were
    offered before the policy ran: the same as

        add_batch({**obs, **acts, **outs}, workers, mask=(tuple(acts), flags, masked))

    (driver.py:72-79: the actions are stored as `value * ~flags` and written to
    `masked[name]`, which is returned; `masked=False`: pool rows only, see
    `carry_publish`).  When the step went in through a step
    record (_StepRecord) and the action / output tensors are ready device
    tensors, the observation keys need no second look: their pointers were
    collected for the early insert."""
    rec = self._cur_rec
    wanted = masked is not False
    if not wanted:
      # Carried publish: the library may leave the action's pool write to its
      # next launch (emb_replay_carry_publish) -- the tensors stay referenced
      # from here until this replay's next step replaces them.
      self._carry_keep = (acts, outs, flags)
    if rec is not None and self._pre_token and obs is self._offer_obs and workers is rec.workers:
      entry = rec.publishes.get(id(masked) if wanted else 0)
      # (the mask outputs are looked at tensor by tensor: a tensor replaced inside
      # the same dict has another address)
      if entry is not None and entry[0] is masked and (not wanted or fast.same_values(masked, entry[5])):
        _, plan, ptrs, mask_plan, flags_at, _, flags_ptr = entry
        extra = {**acts, **outs} if outs else acts
        if (len(extra) == len(plan) and rec.values[flags_at] is flags
            and not fast.columns(extra, plan, ptrs, torch.Tensor, self.device.index)):
          ids, codes, outs_ptr = mask_plan
          with self._lock:
            token, self._pre_token, self._offer_obs, self._cur_rec = self._pre_token, 0, None, None
            while True:
              try:
                fast.emb_replay_publish(
                    self._h, rec.n, rec.workers_ptr, ptrs, len(ids), ids, codes, outs_ptr,
                    flags_ptr, token, self._stream())
                break
              except _lib.PoolFull:
                self._grow(2 * rec.n)
            if self._foreign_selector:
              self._native.reraise()
          return masked if wanted else None
    names = tuple(acts)
    token = self._pre_token
    result = self.add_batch({**obs, **acts, **outs}, workers, mask=(names, flags, masked))
    if (rec is not None and token and fast.columns is not None
        and (result is masked if wanted else result is None)
        and (not wanted or len(masked) == len(names)) and len(rec.publishes) < 16
        and flags.dtype in (torch.bool, torch.uint8) and flags.is_contiguous()
        and flags.device == self.device):
      # What the general path just worked out, kept for the next time this record
      # meets this set of mask outputs: the pointer table (observation keys in
      # place, the actions' slots are refilled per step) and the mask plan.
      flags_at = [i for i, v in enumerate(rec.values) if v is flags]
      extra = {**acts, **outs} if outs else acts
      if flags_at and all(name in self._keyid for name in extra):
        plan = tuple(self._column(self._keyid[name], rec.n, name) for name in extra)
        ptrs = (C.c_void_p * len(self._keys))(*self._batch_ptrs)
        if not fast.columns(extra, plan, ptrs, torch.Tensor, self.device.index):    # every value ready
          ids, codes, _ = self._mask_plans[names]
          if wanted:
            outs_ptr = (C.c_void_p * len(names))(*[masked[name].data_ptr() for name in names])
          else:
            outs_ptr = (C.c_void_p * len(names))()
          # (every entry keeps its own flag address: entries of one record may have
          # been made with different flag tensors)
          rec.publishes[id(masked) if wanted else 0] = (
              masked, plan, ptrs, (ids, codes, outs_ptr), flags_at[0],
              tuple(masked.values()) if wanted else (), flags.data_ptr())
    return result

  def _collect(self, values, columns, ptrs):
    """data_ptr() of every ready value (a contiguous tensor of the column's dtype
    and shape on this GPU) into ptrs[column]; returns the (value, column) pairs
    that are not ready, or None.  A tensor object that passed is marked
    (`_emb_ok` = its column entry): vector envs hand out the same few tensor
    objects step after step, and the next call takes the data_ptr() of a marked
    one after a single identity comparison."""
    if fast.columns is not None:
      todo = fast.columns(values, columns, ptrs, torch.Tensor, self.device.index)
      if todo:
        listed = list(values.values())
        todo = [(listed[at], columns[at]) for at in todo]
      return todo
    todo, index, Tensor = [], self.device.index, torch.Tensor
    for value, column in zip(values.values(), columns):
      i, dtype, shape, name = column[:4]
      if i < 0:
        continue                                    # not stored / not part of this pass
      if type(value) is Tensor and getattr(value, '_emb_ok', None) is column:
        ptrs[i] = value.data_ptr()                  # this tensor object passed the checks before
      elif (type(value) is Tensor and value.dtype is dtype and value.shape == shape
            and value.is_contiguous() and value.get_device() == index):
        value._emb_ok = column
        ptrs[i] = value.data_ptr()
      else:
        todo.append((value, column))
    return todo

  def _column(self, i, n, name):
    """(pool column or -1, dtype, (n, *shape), name): ONE object per (column, n),
    shared by every plan, because `_collect` marks a tensor that passed the checks
    with the column entry it passed them against."""
    if i < 0:
      return (-1, None, None, name)
    spec = self._colspecs.get((i, n))
    if spec is None:
      key = self._keys[i]
      spec = self._colspecs[(i, n)] = (i, key.dtype, (n, *key.shape), name, self.device.index)
    return spec

  def _plan_columns(self, order, n):
    """For every position of a step dict with keys `order`: (pool column or -1,
    dtype, (n, *shape), name).  Raises like the per-step checks did when the key
    set differs from the first step's."""
    columns, seen = [], 0
    for name in order:
      i = self._keyid.get(name)
      if i is None:
        if name.startswith('log/'):
          columns.append(self._column(-1, n, name))
          continue
        raise KeyError(f'replay step key {name!r} was not in the first step')
      columns.append(self._column(i, n, name))
      seen += 1
    if seen + 1 != len(self._keys):
      raise KeyError(f'replay step keys {sorted(order)} differ from the first step')
    return tuple(columns)

  # ----------------------------------------------------------------- sample --

  def sample(self, batch, mode='train', out=None):
    """`batch` sequences of `length` steps -> dict of (batch, length, ...)
    (replay.py:121-127): index draws on the host, one gather launch.

    The returned tensors are the caller's, like the reference's fresh arrays.
    `out=` gathers into tensors the caller already owns instead (a batch an
    earlier `sample` returned, or any dict with this replay's keys as contiguous
    (batch, length, *shape) tensors on its device)."""
    assert mode in _lib.MODES, mode
    if not self._nonempty:
      # (items only leave when a new one pushes them out at capacity: a sampler
      # that has held an item once never runs empty again -- asked once, not with
      # a C call and a clock read per sample)
      limiters.wait(
          lambda: len(self._native), f'Replay buffer {self.name} is empty')
      self._nonempty = True
    with self._lock:
      self._flush()
      stream = self._stream()
      lens = self._key_lens
      if out is None:
        out, ptrs = self._alloc_batch(batch, self.length, lens is not None)
      else:
        out, ptrs = self._adopt(out, batch, self.length, lens is not None)
      # Host copy of stepid[:, 0] rides on the tensor object so `update` with
      # the same tensor needs no device read-back (a sync).  A set that is used
      # again (recycled, `out=`) brings its buffer along.
      sid = out['stepid']
      first = sid.__dict__.get('_emb_first')
      if first is None or len(first) != batch * _lib.STEPID_BYTES:
        first = sid._emb_first = (C.c_uint8 * (batch * _lib.STEPID_BYTES))()
      if lens is None:
        fast.emb_replay_sample(
            self._h, batch, _lib.MODES[mode], ptrs, None, first, stream)
      else:
        fast.emb_replay_sample_heads(
            self._h, batch, _lib.MODES[mode], ptrs, lens, None, first, stream)
      if self._foreign_selector:
        self._native.reraise()
    return self._finish(out)

  def recycle(self, batch):
    """Hand a sampled batch back: its tensors become the output of a later
    `sample` / `gather` of the same shape instead of a fresh allocation (seven
    `torch.empty` per batch are ~9 us of host time).  A promise by the caller:
    nothing reads these tensors any more, except work already queued on the
    stream the batch was sampled on (the next gather is ordered behind it, as
    with the caching allocator handing a freed block out again).  A consumer
    that read the batch on ANOTHER stream orders that stream first.
    `streams.Stateless(replay.sample, B, recycle=K)` does this for batches that
    are K draws old."""
    if type(batch) is not Batch or batch._emb_owner is not self._offer_tag:
      raise TypeError('Replay.recycle takes a batch that this replay\'s sample() / gather() returned')
    sets = self._free.setdefault(batch._emb_shape, [])
    if len(sets) < 8 and all(b is not batch for b in sets):
      sets.append(batch)

  def _adopt(self, out, batch, length, heads=False):
    """`out=`: a Batch of this replay with the right shape is taken as is;
    anything else is checked key by key."""
    shape = (batch, length, 'heads') if heads else (batch, length)
    if type(out) is Batch and out._emb_owner is self._offer_tag and out._emb_shape == shape:
      out._emb_stream = self._last_stream
      return out, out._emb_ptrs
    if self._keys is None or set(out) != {k.name for k in self._keys}:
      raise KeyError(f'sample(out=): keys {sorted(out)} are not this replay\'s')
    ptrs = (C.c_void_p * len(self._keys))()
    for i, key in enumerate(self._keys):
      tensor = out[key.name]
      want = (batch, self._key_lens[i] if heads else length, *key.shape)
      if (not torch.is_tensor(tensor) or tensor.dtype != key.dtype or tuple(tensor.shape) != want
          or tensor.device != self.device or not tensor.is_contiguous()):
        raise ValueError(f'sample(out=): {key.name!r} must be a contiguous {key.dtype} tensor of shape '
                         f'{want} on {self.device}')
      ptrs[i] = tensor.data_ptr()
    adopted = Batch((k.name, out[k.name]) for k in self._keys)
    adopted._emb_ptrs, adopted._emb_shape = ptrs, shape
    adopted._emb_stream, adopted._emb_owner = self._last_stream, self._offer_tag
    return adopted, ptrs

  def _alloc_batch(self, batch, length, heads=False):
    """Output tensors for one sampled batch: (Batch name -> tensor, ctypes array
    of their addresses).  `sample` returns tensors the caller owns, like the
    reference's fresh arrays (replay.py:255-275): fresh allocations, unless

    * the caller handed sets back (`recycle`) -- taken first, same stream only;
    * `Replay(reuse_outputs=K)` rotates K sets (a batch is overwritten by the
      K-th sample after it: whoever holds batches longer -- a Prefetch of
      `amount` batches needs K >= amount + 2 -- must not use it)."""
    shape = (batch, length, 'heads') if heads else (batch, length)
    if self._reuse:
      return self._alloc_batch_now(shape)
    stream = self._last_stream
    sets = self._free.get(shape)
    if sets:
      for i in range(len(sets) - 1, -1, -1):
        if sets[i]._emb_stream == stream:
          out = sets.pop(i)
          return out, out._emb_ptrs
    return self._new_batch(shape)

  def _as_batch(self, tensors, ptrs, shape):
    out = Batch(zip(self._key_names, tensors))
    out._emb_ptrs, out._emb_shape = ptrs, shape
    out._emb_stream, out._emb_owner = self._last_stream, self._offer_tag
    return out

  def _alloc_batch_now(self, shape):
    ring = self._out_ring.setdefault(shape, [[], 0])
    if len(ring[0]) < self._reuse:
      ring[0].append(self._new_batch(shape))
    out, ptrs = ring[0][ring[1] % len(ring[0])]
    ring[1] += 1
    return self._as_batch(list(out.values()), ptrs, shape), ptrs

  def _new_batch(self, shape):
    """shape = (batch, length) or (batch, length, 'heads'): the latter gives the
    keys of `Replay(heads=)` their own, shorter, time axis."""
    # torch.empty_like on a zero-stride, one-element template is about twice as
    # cheap on the host as torch.empty(shape, dtype, device) and yields the same
    # contiguous tensor.
    templates = self._templates.get(shape)
    if templates is None:
      batch, length = shape[:2]
      lens = self._key_lens if len(shape) == 3 else [length] * len(self._keys)
      templates = self._templates[shape] = [
          torch.empty(1, dtype=key.dtype, device=self.device).expand(
              batch, lens[i], *key.shape) for i, key in enumerate(self._keys)]
      self._key_names = [k.name for k in self._keys]
    out, ptrs = Batch(), (C.c_void_p * len(self._keys))()
    for i, key in enumerate(self._keys):
      tensor = out[key.name] = torch.empty_like(templates[i])
      ptrs[i] = tensor.data_ptr()
    out._emb_ptrs, out._emb_shape = ptrs, shape
    out._emb_stream, out._emb_owner = self._last_stream, self._offer_tag
    return out, ptrs

  def _finish(self, out):
    if self.numpy:
      return {k: v.cpu().numpy() for k, v in out.items()}
    return out

  def sample_index(self, batch, mode='train'):
    """Index-only draw: (batch, length) pool rows and the came-from-online-queue
    flags.  Advances the PRNG exactly like `sample`."""
    rows = np.zeros((batch, self.length), np.int32)
    online = np.zeros(batch, np.uint8)
    with self._lock:
      api.emb_replay_sample_index(
          self._handle, batch, _lib.MODES[mode], _lib.ptr(rows), _lib.ptr(online), None)
      self._reraise()
    return rows, online.astype(bool)

  def gather(self, rows):
    """Materialise an explicit (batch, length) row table (owner-side gather in
    the sharded layout; also used when loading)."""
    rows = np.ascontiguousarray(rows, np.int32)
    batch, length = rows.shape
    with self._lock:
      self._flush()
      stream = self._stream()
      out, ptrs = self._alloc_batch(batch, length)
      # (a reused output set: the host copy of stepid[:, 0] that `sample` leaves on
      # the tensor for `update` belongs to an older batch)
      out['stepid'].__dict__.pop('_emb_first', None)
      api.emb_replay_gather_rows(
          self._handle, _lib.ptr(rows), rows.size, length, ptrs, stream)
    return self._finish(out)

  def sample_windows(self, batch, length, consec, prefix=0, mode='train'):
    """`sample` + `Consec` windowing in one pass over the pool
    (replay.py:121-127 + streams.py:120-140): returns `consec` dicts, window w =
    steps [w*length, w*length + length + prefix) of each sampled sequence, each
    key contiguous.  The wide keys are gathered straight into their windows (the
    overlapping prefix rows are read twice from HBM instead of copying every
    window again); only the two 1-byte flag keys take the annotate-then-slice
    route, because the reference annotates the full sequence before slicing."""
    assert mode in _lib.MODES, mode
    assert consec * length + prefix == self.length, (consec, length, prefix, self.length)
    if self._heads:
      raise ValueError('sample_windows: this replay cuts keys to their first steps (heads=), which holds for '
                       'the whole sequence, not for each window: use sample()')
    limiters.wait(
        lambda: len(self._native), f'Replay buffer {self.name} is empty')
    width = length + prefix
    with self._lock:
      self._flush()
      rows, _ = self.sample_index(batch, mode)
      index = (np.arange(consec)[:, None] * length + np.arange(width)[None, :]).reshape(-1)
      tiled = np.ascontiguousarray(
          rows[:, index].reshape(batch, consec, width).transpose(1, 0, 2))
      flags = [k for k in ('is_first', 'is_last') if k in self._keyid]
      out, wide, narrow = {}, (C.c_void_p * len(self._keys))(), (C.c_void_p * len(self._keys))()
      full = {}
      for i, key in enumerate(self._keys):
        if key.name in flags:
          full[key.name] = torch.empty(
              (batch, self.length, *key.shape), dtype=key.dtype, device=self.device)
          narrow[i] = full[key.name].data_ptr()
        else:
          out[key.name] = torch.empty(
              (consec, batch, width, *key.shape), dtype=key.dtype, device=self.device)
          wide[i] = out[key.name].data_ptr()
      stream = self._stream()
      api.emb_replay_gather_rows(
          self._handle, _lib.ptr(tiled), tiled.size, width, wide, stream)
      if flags:
        api.emb_replay_gather_rows(
            self._handle, _lib.ptr(rows), rows.size, self.length, narrow, stream)
    windows = []
    for w in range(consec):
      win = {}
      for key in self._keys:
        if key.name in full:
          win[key.name] = full[key.name][:, w * length: w * length + width].contiguous()
        else:
          win[key.name] = out[key.name][w]
      windows.append(self._finish(win))
    return windows

  # ----------------------------------------------------------------- update --

  def update(self, data):
    """Write agent outputs back over sampled steps and/or re-prioritise
    (replay.py:129-149).  Rows whose first chunk was evicted are skipped."""
    data = dict(data)
    stepid = data.pop('stepid')
    priority = data.pop('priority', None)
    assert stepid.ndim == 3, stepid.shape
    first = getattr(stepid, '_emb_first', None)
    steps = int(np.prod(stepid.shape[:-1]))
    if first is None or priority is not None:
      if torch.is_tensor(stepid):
        stepid = stepid.detach().cpu().numpy()
      stepid = np.ascontiguousarray(stepid, np.uint8)
      first = np.ascontiguousarray(stepid[:, 0])
    if isinstance(first, np.ndarray):
      B, first_ptr = len(first), _lib.ptr(first)
    else:                                  # the ctypes buffer `sample` attached
      B, first_ptr = len(first) // _lib.STEPID_BYTES, first
    with self._lock:
      self._flush()
      if priority is not None:
        if torch.is_tensor(priority):
          priority = priority.detach().cpu().numpy()
        assert np.ndim(priority) == 2, np.shape(priority)
        prios = np.ascontiguousarray(priority, np.float64).reshape(-1)
        flat = stepid.reshape(-1, stepid.shape[-1])
        if not hasattr(self.sampler, 'prioritize'):
          raise AttributeError(       # what replay.py:137 raises
              f'{type(self.sampler).__name__!r} object has no attribute '
              "'prioritize'")
        api.emb_replay_prioritize(
            self._handle, _lib.ptr(flat), _lib.ptr(prios), len(prios))
        self._reraise()
      if data:
        T = None
        ids = (C.c_int32 * len(data))()
        ptrs = (C.c_void_p * len(data))()
        keep = []
        for j, (name, value) in enumerate(data.items()):
          key = self._keys[self._keyid[name]]
          if not torch.is_tensor(value):
            value = torch.from_numpy(np.ascontiguousarray(value))
          value = value.to(self.device, key.dtype, non_blocking=True).contiguous()
          T = value.shape[1] if T is None else T
          if tuple(value.shape) != (B, T, *key.shape):
            raise ValueError((name, tuple(value.shape), (B, T, *key.shape)))
          keep.append(value)
          ids[j] = self._keyid[name]
          ptrs[j] = value.data_ptr()
        fast.emb_replay_update(
            self._h, B, T, first_ptr, len(data), ids, ptrs, self._stream())
      # replay.py:134: every call counts B*T steps, written or not.
      self._updates += steps

  def _reraise(self):
    if self._foreign_selector:
      self._native.reraise()

  # ------------------------------------------------------------ save / load --

  def _no_sharded_checkpoint(self, what):
    if self._owners > 1:
      raise NotImplementedError(
          f'Replay.{what}: a sharded pool (owners={self._owners}) keeps only one owner\'s rows '
          'on this rank; checkpoint each rank\'s own Replay instead')

  def _reserve_directory_uids(self, directory):
    """Never reissue the id of a chunk file that is already in `directory`
    (a fresh Replay pointed at a used directory, a skipped corrupted file, train
    and eval replays sharing a directory): move the serial past all of them."""
    top = 0
    for path in pathlib.Path(directory).glob('*.npz'):
      try:
        _, uid, succ, _ = parse_filename(path.name)
      except Exception:
        continue
      # The successor named in a file name may never have been written (it was
      # still empty): its id is taken all the same.
      for taken in (uid, succ):
        if taken >> 64 == self._replica:
          top = max(top, taken & _MASK64)
    if top:
      api.emb_replay_reserve_uids(self._handle, top + 1)

  def _chunk_table(self):
    n = C.c_int64()
    api.emb_replay_chunks(self._handle, 0, None, None, None, None, None, C.byref(n))
    cap = n.value
    uid, succ = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
    fill, slot, tms = (np.zeros(cap, np.int64) for _ in range(3))
    api.emb_replay_chunks(
        self._handle, cap, _lib.ptr(uid), _lib.ptr(succ), _lib.ptr(fill),
        _lib.ptr(slot), _lib.ptr(tms), C.byref(n))
    return [
        dict(uid=int(uid[i]), succ=int(succ[i]), fill=int(fill[i]),
             slot=int(slot[i]), time_ms=int(tms[i])) for i in range(min(cap, n.value))]

  def _full_uid(self, serial):
    return (self._replica << 64) | serial if serial else 0

  def save(self):
    """Write every unsaved, non-empty chunk as `{time}-{uid}-{succ}-{length}.npz`
    (chunk.py:31-33,64-75; replay.py:294-309): open chunks are closed first, the
    rows come back from HBM with one copy per key.  Returns None: the directory
    is the state."""
    if not self.directory:
      return None
    self._no_sharded_checkpoint('save')
    directory = pathlib.Path(self.directory)
    directory.mkdir(parents=True, exist_ok=True)
    with self._lock:
      self._flush()
      self._reserve_directory_uids(directory)
      # Closing the open chunks opens one successor per worker: make room first
      # (the pool grows lazily, a checkpoint must not die of PoolFull).
      while True:
        need, free = C.c_int64(), C.c_int64()
        api.emb_replay_open_chunks(self._handle, C.byref(need))
        api.emb_replay_free_slots(self._handle, C.byref(free))
        if free.value < need.value:
          self._grow(need.value)
        try:
          api.emb_replay_complete_all(self._handle)    # all or nothing
          break
        except _lib.PoolFull:
          self._grow(need.value)
      jobs = []
      for chunk in self._chunk_table():
        if chunk['fill'] <= 0 or chunk['uid'] in self._saved:
          continue
        self._saved.add(chunk['uid'])
        lo = chunk['slot'] * self.chunksize
        data = {}
        for key in self._keys:
          rows = key.pool[lo * key.rowbytes: (lo + chunk['fill']) * key.rowbytes]
          host = rows.cpu().numpy()
          if key.dtype == torch.bfloat16:
            host = host.view(np.uint16)
          else:
            host = host.view(_numpy_of(key.dtype))
          data[key.name] = host.reshape(chunk['fill'], *key.shape)
        name = chunk_filename(
            chunk['time_ms'], self._full_uid(chunk['uid']),
            self._full_uid(chunk['succ']), chunk['fill'])
        jobs.append(self._savers.submit(_write_npz, directory / name, data))
      if self.save_wait:
        [job.result() for job in jobs]
    return None

  def load(self, data=None, directory=None, amount=None):
    """Restore the newest chunks from disk until `amount` items are back
    (replay.py:311-359): file order, per-chunk item counts and reference
    counting as the reference; payload goes up with one copy per key per chunk."""
    directory = _as_path(directory) or self.directory
    amount = amount or self.capacity or np.inf
    if not directory:
      return
    self._no_sharded_checkpoint('load')
    directory = pathlib.Path(directory)
    with self._lock:
      self._flush()
      table = self._chunk_table()
      loaded = sorted((chunk_filename(
          c['time_ms'], self._full_uid(c['uid']), self._full_uid(c['succ']), c['fill'])
          for c in table), reverse=True)
      loaded_uids = {self._full_uid(c['uid']) for c in table}
      ondisk = sorted((p.name for p in directory.glob('*.npz')), reverse=True)
      ondisk = [x for x in ondisk if parse_filename(x)[1] not in loaded_uids]
      # Chunk ids are `replica << 64 | serial` here; chunk.py:15-16 draws random
      # 128-bit UUIDs, and another replica's files carry its id in the upper
      # half.  The reference's load() takes every file in the directory whoever
      # wrote it (replay.py:311-359), so such files are loaded too, under fresh
      # local serials: the uid -> succ links are kept, the step ids inside are
      # re-issued (stepid[:16] names the chunk) and a file is never taken twice.
      live = {c['uid'] for c in table}
      ondisk = [x for x in ondisk
                if self._foreign.get(parse_filename(x)[1], -1) not in live]
      foreign = [x for x in ondisk if parse_filename(x)[1] >> 64 != self._replica]
      if foreign:
        top = max([c['uid'] for c in table] + [
            u & _MASK64 for x in ondisk for u in parse_filename(x)[1:3]
            if u >> 64 == self._replica] + list(self._foreign.values()) + [0])
        for name in reversed(foreign):                     # oldest first
          for uid in parse_filename(name)[1:3]:
            if uid and uid >> 64 != self._replica and uid not in self._foreign:
              top += 1
              self._foreign[uid] = top
        api.emb_replay_reserve_uids(self._handle, top + 1)
        print(f'Loading {len(foreign)} chunk file(s) written under other chunk ids '
              f'(reference UUIDs or another replica) as local chunks: {foreign[0]} ...')
      if not ondisk:
        return
      counts = count_items(loaded + ondisk, self.length)
      total, take = 0, 0
      for name in ondisk:
        take += 1
        total += counts[parse_filename(name)[1]]
        if total >= amount:
          break
      chunks, taken = [], set()
      for name in ondisk[:take]:
        uid = parse_filename(name)[1]
        if uid in taken:                  # the same chunk saved twice: keep the newest file
          print(f'Skipping duplicate chunk file {name}')
          continue
        taken.add(uid)
        try:
          with np.load(directory / name) as f:
            arrays = {k: f[k] for k in f.files}
        except Exception as e:            # corrupted file: skip (chunk.py:81-91)
          print(f'Error loading chunk {name}: {e}')
          continue
        chunks.append((name, arrays))
      if not chunks:
        return
      counts = count_items([name for name, _ in chunks], self.length)
      if self._keys is None:
        first = {k: v[0] for k, v in chunks[0][1].items() if k != 'stepid'}
        self._init_keys(first)
      free = C.c_int64()
      api.emb_replay_free_slots(self._handle, C.byref(free))
      if free.value < len(chunks) + 2:
        self._grow(len(chunks))
      for name, arrays in chunks:
        time_ms, uid, succ, length = parse_filename(name)
        slot = C.c_int64()
        api.emb_replay_load_chunk(
            self._handle, self._local_uid(uid), self._local_uid(succ), length, time_ms,
            C.byref(slot))
        lo = slot.value * self.chunksize
        if uid >> 64 != self._replica:
          # re-issue the step ids: 16-byte big-endian chunk id | 4-byte row
          ids = np.array(arrays['stepid'][:length], np.uint8)
          ids[:, :16] = np.frombuffer(
              ((self._replica << 64) | self._local_uid(uid)).to_bytes(16, 'big'), np.uint8)
          arrays = {**arrays, 'stepid': ids}
        for key in self._keys:
          host = np.ascontiguousarray(arrays[key.name][:length])
          flat = torch.from_numpy(host.reshape(-1).view(np.uint8))
          key.pool[lo * key.rowbytes: (lo + length) * key.rowbytes].copy_(flat)
        self._saved.add(self._local_uid(uid))
      for name, _ in reversed(chunks):
        _, uid, _, _ = parse_filename(name)
        api.emb_replay_load_items(self._handle, self._local_uid(uid), int(counts[uid]))
      self._reraise()

  def _local_uid(self, uid):
    """The 64-bit chunk serial this replay uses for a chunk id found in a file
    name: its lower half if this replica issued it, else the serial `load`
    assigned to the foreign id."""
    if uid == 0 or uid >> 64 == self._replica:
      return uid & _MASK64
    return self._foreign[uid]

  # ------------------------------------------------------------- profiling --

  def profile(self, enable=True, every=1):
    """HIP-event timing of this replay's gather launches (bench roofline);
    `every=n` stamps one gather in n."""
    api.emb_replay_profile(self._handle, int(every) if enable and every > 1 else int(bool(enable)))

  def profile_read(self, reset=True):
    launches, ms = C.c_int64(), C.c_double()
    api.emb_replay_profile_read(
        self._handle, C.byref(launches), C.byref(ms), int(reset))
    return launches.value, ms.value

  def profile_report(self, which='sample', reset=True):
    """(stamped launches, their total ms, kernel name) of the sample gathers or
    of the `update` write-backs since the last reset; 'deferred': how many
    publishes handed their index bookkeeping to the library's helper thread;
    'carried': (carried publishes that rode in the next early-insert launch, all
    carried publishes -- in the `ms` position)."""
    launches, ms, name = C.c_int64(), C.c_double(), C.create_string_buffer(128)
    api.emb_replay_profile_report(
        self._handle, {'sample': 0, 'update': 1, 'deferred': 2, 'carried': 3}[which], C.byref(launches), C.byref(ms),
        int(reset), name, len(name))
    return launches.value, ms.value, name.value.decode()


def _as_path(directory):
  """None / '' stay falsy; str, os.PathLike or anything whose str() is the path
  (elements.Path, ppo/main.py:190) -> pathlib.Path."""
  if not directory:
    return None
  return pathlib.Path(directory if isinstance(directory, (str, os.PathLike)) else str(directory))


def _all_host_arrays(steps):
  ndarray = np.ndarray
  for value in steps.values():
    if type(value) is not ndarray:
      return False
  return True


def _numpy_of(dtype):
  for np_dtype, t in _TORCH_OF.items():
    if t == dtype:
      return np_dtype
  raise TypeError(dtype)


_MASK64 = (1 << 64) - 1
_B62 = '0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'


def _b62(value):
  chars = []
  while value:
    value, rem = divmod(value, 62)
    chars.append(_B62[rem])
  return ''.join(reversed(chars)).rjust(22, '0')


def _unb62(text):
  value = 0
  for char in text:
    value = value * 62 + _B62.index(char)
  return value


def chunk_filename(time_ms, uid, succ, length):
  """`{time}-{uid}-{succ}-{length}.npz` (chunk.py:31-33): sortable timestamp
  with milliseconds, 128-bit ids in base 62."""
  stamp = time.strftime('%Y%m%dT%H%M%S', time.gmtime(time_ms / 1000))
  return f'{stamp}F{time_ms % 1000:03d}-{_b62(uid)}-{_b62(succ)}-{length}.npz'


def parse_filename(name):
  stem = name[:-4] if name.endswith('.npz') else name
  stamp, uid, succ, length = stem.split('-')
  try:
    import calendar
    secs = calendar.timegm(time.strptime(stamp[:15], '%Y%m%dT%H%M%S'))
    time_ms = secs * 1000 + int(stamp[16:19] or 0)
  except Exception:
    time_ms = 0
  return time_ms, _unb62(uid), _unb62(succ), int(length)


def count_items(names, length):
  """Items each chunk file contributes once its successors are present
  (replay.py:372-388)."""
  stems = sorted((n[:-4] if n.endswith('.npz') else n for n in names), reverse=True)
  parsed = [parse_filename(s) for s in stems]
  lengths = {uid: n for _, uid, _, n in parsed}
  future = {}
  for _, uid, succ, n in parsed:
    future[uid] = n + future.get(succ, 0)
  counts = {}
  for _, uid, succ, n in parsed:
    counts[uid] = int(np.clip(n + 1 - length + future.get(succ, 0), 0, lengths[uid]))
  return counts


def _write_npz(path, data):
  with io.BytesIO() as stream:
    np.savez_compressed(stream, **data)
    path.write_bytes(stream.getvalue())
