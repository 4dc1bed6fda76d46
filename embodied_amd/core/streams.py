"""Batch streams (reference: embodied/core/streams.py): `Stateless` wraps a
sampling function, `Consec` serves one long sampled batch as consecutive
overlapping windows, `Prefetch` runs a source some batches ahead in a thread.

Batches may be dicts of numpy arrays or of torch device tensors; for device
tensors the window copy is one `emb_window_keys` launch for all keys, and over
this package's `Replay.sample` the sampling and the windowing are one gather.
"""
import collections
import functools
import threading

import numpy as np
import torch

from .. import _lib
from .._lib import api
from . import base


class Stateless(base.Stream):
  """An endless stream that calls `fn(*args, **kwargs)` for every batch; an
  iterator may be given instead of a function (streams.py:12-29).  It carries
  no state of its own: `save()` is None and `load` ignores its argument.

  `recycle=K` (this package's `Replay.sample` only; not in the reference, whose
  batches are fresh arrays): the stream lends its batches instead of giving them
  away -- a batch belongs to the stream again once K further batches have been
  drawn and is then handed back to the replay (`Replay.recycle`), which gathers
  a later batch into the same tensors instead of allocating seven new ones.  The
  consumer of such a stream keeps no batch (and no view of one) for longer than
  K draws, and reads it on the stream it was sampled on.  (`Consec(consec > 1)`
  over such a stream samples through `Replay.sample_windows` and lends nothing:
  `recycle` only concerns whole-batch draws.)"""

  def __init__(self, fn, *args, recycle=0, **kwargs):
    if not callable(fn):
      if not hasattr(fn, '__next__'):
        raise TypeError(f'Stateless needs a callable or an iterator, got {type(fn).__name__}')
      fn = fn.__next__
    if type(fn) is functools.partial:
      # `Stateless(bind(replay.sample, batch, mode))` (ppo/main.py:262-263) is
      # `Stateless(replay.sample, batch, mode)`: unwrapped, so that `recycle` and
      # Consec's fused route see the Replay behind it.
      fn, args, kwargs = fn.func, (*fn.args, *args), {**fn.keywords, **kwargs}
    self.fn, self.args, self.kwargs = fn, args, kwargs
    self.recycle = int(recycle)
    self._lent = collections.deque()
    self._give_back = None
    if self.recycle:
      owner = getattr(fn, '__self__', None)
      self._give_back = getattr(owner, 'recycle', None)
      if self.recycle < 0 or self._give_back is None or getattr(fn, '__name__', '') != 'sample':
        raise TypeError('Stateless(recycle=K) needs K >= 1 and a Replay.sample of this package as `fn`')
      if getattr(owner, 'numpy', False):
        raise TypeError('Stateless(recycle=K): Replay(numpy=True) returns host copies, there is nothing '
                        'to hand back')

  def __iter__(self):
    return self

  def __next__(self):
    if not self.recycle:
      return self.fn(*self.args, **self.kwargs)
    lent = self._lent
    if len(lent) > self.recycle:
      self._give_back(lent.popleft())
    batch = self.fn(*self.args, **self.kwargs)
    lent.append(batch)
    return batch

  def save(self):
    return None

  def load(self, data):
    del data


def window(value, start, count):
  """value[:, start:start+count] as a contiguous array (streams.py:133-138)."""
  if not torch.is_tensor(value):
    return np.ascontiguousarray(value[:, start: start + count])
  if start == 0 and count == value.shape[1]:
    return value if value.is_contiguous() else value.contiguous()
  if not value.is_cuda:
    return value[:, start: start + count].contiguous()
  value = value.contiguous()
  out = torch.empty(
      (value.shape[0], count, *value.shape[2:]), dtype=value.dtype,
      device=value.device)
  rowbytes = value.element_size() * int(np.prod(value.shape[2:], dtype=np.int64))
  if out.numel() == 0:
    return out
  api.emb_window(
      value.data_ptr(), out.data_ptr(), value.shape[0], value.shape[1], start,
      count, rowbytes, _lib.raw_stream(value.device))
  return out


def window_batch(batch, start, count):
  """`{k: v[:, start:start+count]}` for a dict of device tensors with ONE kernel
  launch for all keys (`emb_window_keys`)."""
  import ctypes as C
  names = list(batch)
  first = batch[names[0]]
  total = first.shape[1]
  # One (batch, total) addresses every key inside the kernel: a key of another
  # length (a context-only key of `Replay(heads=)`, (B, K, ...)) would be read
  # out of bounds or from the wrong sequences.
  odd = [k for k in names if batch[k].shape[:2] != first.shape[:2]]
  if odd:
    raise ValueError(f'window_batch: keys {odd} are not shaped ({first.shape[0]}, {total}, ...) like '
                     f'{names[0]!r} (context-only keys cannot be windowed)')
  if start == 0 and count == total:
    return {k: (v if v.is_contiguous() else v.contiguous()) for k, v in batch.items()}
  srcs = [batch[k] if batch[k].is_contiguous() else batch[k].contiguous() for k in names]
  outs = [torch.empty((v.shape[0], count, *v.shape[2:]), dtype=v.dtype, device=v.device)
          for v in srcs]
  n = len(names)
  rowbytes = (C.c_int64 * n)(*[
      v.element_size() * int(np.prod(v.shape[2:], dtype=np.int64)) for v in srcs])
  api.emb_window_keys(
      n, (C.c_void_p * n)(*[v.data_ptr() for v in srcs]),
      (C.c_void_p * n)(*[v.data_ptr() for v in outs]), rowbytes, first.shape[0],
      total, start, count, _lib.raw_stream(first.device))
  return dict(zip(names, outs))


_BATCH_TYPE = []


def _batch_type():
  if not _BATCH_TYPE:
    from . import replay as replaylib
    _BATCH_TYPE.append(replaylib.Batch)
  return _BATCH_TYPE[0]


class Consec(base.Stream):
  """Sequence windowing (streams.py:89-150): a source batch of
  `consec * length + prefix` steps is served as `consec` windows
  `[:, i*length : i*length + length + prefix]`, each with an int32 `consec` key
  holding the window number; the prefix columns of successive windows overlap.

  State (`save`/`load`): the source's state and the number of the next window.
  """

  def __init__(
      self, source, length, consec, prefix=0, strict=True, contiguous=False):
    self.source = source
    self.length, self.consec, self.prefix = length, consec, prefix
    self.strict = strict
    self.contiguous = contiguous
    self.index = 0            # next window of the current source batch
    self.current = None       # the source batch being served (unfused route)
    self.windows = None       # its windows, already cut (fused route)
    self._batches = None
    self._fused = False       # not probed yet
    self._numbers = {}        # (shape, window number, device) -> constant tensor

  def __iter__(self):
    self._batches = iter(self.source)
    return self

  def __next__(self):
    number = 0 if self.index >= self.consec else self.index
    fused = self._fused_source() if self.consec > 1 else None
    if fused is not None:
      chunk = self._next_fused(number, *fused)
    else:
      chunk = self._next_sliced(number)
    self.index = number + 1
    return chunk

  def _next_fused(self, number, replay, batch, mode):
    if number == 0:
      self.windows = replay.sample_windows(
          batch, self.length, self.consec, self.prefix, mode)
    chunk = dict(self.windows[number])
    chunk['consec'] = self._number(chunk['is_first'], number)
    return chunk

  def _next_sliced(self, number):
    if number == 0:
      self.current = next(self._batches)
      have = self.current['is_first'].shape[1]
      need = self.length * self.consec + self.prefix
      if have < need or (self.strict and have != need):
        raise AssertionError(
            f'Consec(length={self.length}, consec={self.consec}, prefix={self.prefix}) needs '
            f'{"exactly" if self.strict else "at least"} {need} steps per sequence, got {have}')
      if self.consec > 1 or have != need:
        # whenever a window is not the whole batch -- consec > 1, or consec == 1 cutting
        # `need` of `have` steps (strict=False) -- every key must span the batch
        short = [k for k, v in self.current.items() if v.shape[1] != have]
        if short:       # `Replay(heads=)` keys hold the head of the WHOLE sequence, not of each window
          raise AssertionError(f'Consec(consec={self.consec}): keys {short} do not span the {have} steps '
                               'of the source batch (context-only keys cannot be windowed)')
    start, count = number * self.length, self.length + self.prefix
    batch = self.current
    if (self.consec == 1 and type(batch) is _batch_type()
        and getattr(batch, '_emb_shape', (0, 0))[1] == count):
      # The whole of a batch this package's Replay.sample made (contiguous device
      # tensors by construction): the window is the batch itself -- the same
      # tensors, as the general route below also returns for a whole-batch window.
      # They are the caller's unless the replay rotates its outputs
      # (`Replay(reuse_outputs=K)`, whose contract Prefetch checks) or the source
      # lends them (`Stateless(recycle=K)`).
      chunk = dict(batch)
      chunk['consec'] = self._number(chunk['is_first'], 0)
      return chunk
    if torch.is_tensor(batch['is_first']):
      # Device batches are always materialised contiguously (one kernel for all
      # keys); `contiguous` only matters for numpy views.
      if all(torch.is_tensor(v) and v.is_cuda for v in batch.values()):
        chunk = window_batch(batch, start, count)
      else:
        chunk = {k: window(v, start, count) for k, v in batch.items()}
    else:
      chunk = {k: v[:, start: start + count] for k, v in batch.items()}
      if self.contiguous:
        chunk = {k: np.ascontiguousarray(v) for k, v in chunk.items()}
    chunk['consec'] = self._number(chunk['is_first'], number)
    return chunk

  def _number(self, like, number):
    """The `consec` key: int32, shaped like `is_first`, filled with the window
    number.  numpy batches get a fresh array; device batches share one constant
    tensor per (shape, number) — filling a new one is a kernel launch and ~5 us
    of host time per train step (treat it as read-only, like any batch key the
    agent does not own)."""
    if not torch.is_tensor(like):
      return np.full(like.shape, number, np.int32)
    key = (tuple(like.shape), number, like.device)
    const = self._numbers.get(key)
    if const is None:
      if len(self._numbers) > 64:
        self._numbers.clear()
      const = self._numbers[key] = torch.full(
          like.shape, number, dtype=torch.int32, device=like.device)
    return const

  def _fused_source(self):
    """(replay, batch, mode) if the source is `Stateless(replay.sample, batch,
    mode)` over this package's Replay with matching lengths: then sampling and
    windowing run as one gather (`Replay.sample_windows`)."""
    if self._fused is not False:
      return self._fused
    self._fused = None
    from . import replay as replaylib
    src = self.source
    fn = getattr(src, 'fn', None)
    target = getattr(fn, '__self__', None)
    if (isinstance(src, Stateless) and isinstance(target, replaylib.Replay)
        and getattr(fn, '__func__', None) is replaylib.Replay.sample
        and not src.kwargs and 1 <= len(src.args) <= 2
        and 'is_first' in (target._keyid or {'is_first': 0})
        and not target._heads        # (context-only keys: the sliced route says why they cannot be windowed)
        and target.length == self.consec * self.length + self.prefix):
      mode = src.args[1] if len(src.args) > 1 else 'train'
      self._fused = (target, src.args[0], mode)
    return self._fused

  def save(self):
    return {'source': self.source.save(), 'index': self.index}

  def load(self, data):
    self.source.load(data['source'])
    self.index = data['index']


class _Failed:
  """What the producer thread leaves in the buffer when the source raised."""

  def __init__(self, error):
    self.error = error


class Prefetch(base.Stream):
  """Keeps up to `amount` transformed batches ready ahead of the consumer
  (streams.py:32-86; the agent's `stream()` uses it to overlap batch
  preparation with the train step).

  A producer thread fills a buffer while it holds fewer than `amount` batches.
  Every batch travels with the source's state taken right after it was drawn;
  `save()` returns the state that belongs to the last batch HANDED OUT, so a
  restore replays exactly the batches the consumer has not seen.  `load(state)`
  drops whatever was prepared ahead, lets a batch that is being prepared finish
  (it is dropped as well), restores the source and resumes.  An exception in
  the source or the transform reaches the consumer as RuntimeError on the
  `next()` that would have returned that batch.
  """

  def __init__(self, source, transform=None, amount=1):
    self.amount = int(amount)
    # A source that lends its batches (Stateless(recycle=K)) takes one back K
    # draws later; this stream holds `amount` batches ready and the consumer one
    # more while the producer draws the next.
    inner = source
    while inner is not None and not isinstance(inner, Stateless):
      inner = getattr(inner, 'source', None)
    if inner is not None and 0 < inner.recycle < self.amount + 2:
      raise ValueError(f'Prefetch(amount={self.amount}) over Stateless(recycle={inner.recycle}): '
                       f'needs recycle >= amount + 2 = {self.amount + 2}')
    # The same for a replay that rotates K output sets itself: the batches of
    # such a replay alias one of the K sets all the way through Consec.
    reuse = getattr(getattr(getattr(inner, 'fn', None), '__self__', None), '_reuse', 0)
    if 0 < reuse < self.amount + 2:
      raise ValueError(f'Prefetch(amount={self.amount}) over Replay(reuse_outputs={reuse}): a batch is '
                       f'overwritten {reuse} samples later; needs reuse_outputs >= amount + 2 = '
                       f'{self.amount + 2}')
    self.source = iter(source) if hasattr(source, '__iter__') else source()
    self._transform = transform
    self._state = self._snapshot()
    self._cond = threading.Condition()
    self._ready = collections.deque()
    self._busy = False       # the producer is drawing / transforming a batch
    self._epoch = 0          # bumped by load(): batches of older epochs are dropped
    self._thread = None

  def __iter__(self):
    if self._thread is not None:
      raise AssertionError('Prefetch can be iterated once')
    self._thread = threading.Thread(target=self._produce, name='prefetch', daemon=True)
    self._thread.start()
    return self

  def __next__(self):
    if self._thread is None:
      raise AssertionError('call iter() on a Prefetch before next()')
    with self._cond:
      while not self._ready:
        self._cond.wait()
      item = self._ready.popleft()
      if isinstance(item, _Failed):
        self._ready.appendleft(item)          # stays failed
        raise RuntimeError(str(item.error)) from item.error
      self._cond.notify_all()
    data, self._state = item
    return data

  def save(self):
    return self._state

  def load(self, state):
    with self._cond:
      self._epoch += 1
      self._ready.clear()
      while self._busy:
        self._cond.wait()
      self._ready.clear()
      self.source.load(state)
      self._state = self._snapshot()
      self._cond.notify_all()

  def _produce(self):
    while True:
      with self._cond:
        while len(self._ready) >= self.amount:
          self._cond.wait()
        epoch, self._busy = self._epoch, True
      try:
        data = next(self.source)
        if self._transform is not None:
          data = self._transform(data)
        item = (data, self._snapshot())
      except BaseException as error:      # incl. StopIteration: the consumer must hear of it
        item = _Failed(error)
      with self._cond:
        self._busy = False
        if epoch == self._epoch:
          self._ready.append(item)
        self._cond.notify_all()
        if isinstance(item, _Failed) and epoch == self._epoch:
          return

  def _snapshot(self):
    save = getattr(self.source, 'save', None)
    return save() if save is not None else None


# ---- combinators (streams.py:153-243) ---------------------------------------------
# Not on the hot path; kept so that code composing streams by these names runs.


def _concat(parts):
  """Leaf-wise concatenation along the batch axis of equally structured batches
  (dicts, possibly nested; numpy arrays or torch tensors)."""
  first = parts[0]
  if isinstance(first, dict):
    return type(first)((key, _concat([part[key] for part in parts])) for key in first)
  if torch.is_tensor(first):
    return torch.cat(parts)
  return np.concatenate(parts)


class Zip(base.Stream):
  """One batch per source and step, concatenated leaf by leaf along the batch
  axis (streams.py:153-177).  Checkpoint: the list of the sources' states."""

  def __init__(self, sources):
    if len(sources) < 2:
      raise ValueError(f'Zip needs at least two sources, got {len(sources)}')
    self.sources = list(sources)
    self.iterators = None

  def __iter__(self):
    if self.iterators is not None:
      raise RuntimeError('Zip was already started')
    self.iterators = [iter(source) for source in self.sources]
    return self

  def __next__(self):
    return _concat([next(it) for it in self.iterators])

  def save(self):
    return [it.save() for it in self.iterators]

  def load(self, data):
    if len(data) != len(self.iterators):
      raise ValueError(f'Zip.load: {len(data)} states for {len(self.iterators)} sources')
    for it, state in zip(self.iterators, data):
      it.load(state)


class Map(base.Stream):
  """`fn(batch, *args, **kwargs)` of every batch of `source` (streams.py:180-201);
  the checkpoint is the source's."""

  def __init__(self, source, fn, *args, **kwargs):
    self.source = source
    self.fn, self.args, self.kwargs = fn, args, kwargs
    self.iterator = None

  def __iter__(self):
    if self.iterator is not None:
      raise RuntimeError('Map was already started')
    self.iterator = iter(self.source)
    return self

  def __next__(self):
    if self.iterator is None:
      raise RuntimeError('Map: iter() first')
    return self.fn(next(self.iterator), *self.args, **self.kwargs)

  def save(self):
    return self.iterator.save()

  def load(self, data):
    self.iterator.load(data)


class Mixer(base.Stream):
  """Every step draws ONE of the named sources with probability proportional to
  its weight, from `default_rng([seed, step])`, and returns that source's next
  batch (streams.py:204-243).  Upstream cannot run as written (`np.ranodm`, an
  assert on a flag that is never set, `load` indexing a list by key); with the
  one misspelt token repaired and the flag set its draws are these, source for
  source (tests/test_limiters_streams_host.py, build container)."""

  def __init__(self, sources, weights, seed=0):
    if sources.keys() != weights.keys():
      raise ValueError(f'Mixer: sources {sorted(sources)} and weights {sorted(weights)} differ')
    self.keys = sorted(sources.keys())
    self.iterators = [iter(sources[key]) for key in self.keys]
    weights = np.array([weights[key] for key in self.keys], np.float32)
    self.probs = weights / weights.sum()
    self.seed = seed
    self.step = 0

  def __iter__(self):
    return self

  def __next__(self):
    rng = np.random.default_rng(seed=[self.seed, self.step])
    self.step += 1
    return next(self.iterators[int(rng.choice(len(self.keys), p=self.probs))])

  def save(self):
    return {
        'step': self.step, 'seed': self.seed,
        'sources': {key: it.save() for key, it in zip(self.keys, self.iterators)}}

  def load(self, data):
    if sorted(data['sources'].keys()) != self.keys:
      raise ValueError(f'Mixer.load: sources {sorted(data["sources"])} are not {self.keys}')
    self.step, self.seed = data['step'], data['seed']
    for key, it in zip(self.keys, self.iterators):
      it.load(data['sources'][key])
