"""Batch streams (reference: embodied/core/streams.py): `Stateless` wraps a
sampling function, `Consec` serves one long sampled batch as consecutive
overlapping windows, `Prefetch` runs a source one batch ahead in a thread.

Batches may be dicts of numpy arrays or of torch device tensors; for device
tensors the window copy is the `emb_window` kernel (one launch per key).
"""
import functools
import queue
import threading

import numpy as np
import torch

from .. import _lib
from .._lib import api
from . import base


class Stateless(base.Stream):
  """streams.py:12-29."""

  def __init__(self, nextfn, *args, **kwargs):
    if not callable(nextfn) and hasattr(nextfn, '__next__'):
      nextfn = nextfn.__next__
    self.nextfn = functools.partial(nextfn, *args, **kwargs)

  def __iter__(self):
    return self

  def __next__(self):
    return self.nextfn()

  def save(self):
    return None

  def load(self, data):
    pass


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


class Consec(base.Stream):
  """Sequence windowing (streams.py:89-150): a source batch of
  `consec * length + prefix` steps is served as `consec` windows
  `[:, i*length : i*length + length + prefix]`, each with an int32 `consec` key
  holding the window number; the prefix columns of successive windows overlap."""

  def __init__(
      self, source, length, consec, prefix=0, strict=True, contiguous=False):
    self.source = source
    self.length = length
    self.consec = consec
    self.prefix = prefix
    self.strict = strict
    self.contiguous = contiguous
    self.index = 0
    self.current = None
    self.it = None
    self._fused = False
    self.windows = None

  def __iter__(self):
    self.it = iter(self.source)
    return self

  def __next__(self):
    if self.index >= self.consec:
      self.index = 0
    fused = self._fused_source() if self.consec > 1 else None
    if fused is not None:
      if self.index == 0:
        replay, batch, mode = fused
        self.windows = replay.sample_windows(
            batch, self.length, self.consec, self.prefix, mode)
      chunk = dict(self.windows[self.index])
      first = chunk['is_first']
      if torch.is_tensor(first):
        chunk['consec'] = torch.full(
            first.shape, self.index, dtype=torch.int32, device=first.device)
      else:
        chunk['consec'] = np.full(first.shape, self.index, np.int32)
      self.index += 1
      return chunk
    if self.index == 0:
      self.current = next(self.it)
      have = self.current['is_first'].shape[1]
      need = self.length * self.consec + self.prefix
      assert need <= have, (self.length, self.consec, self.prefix, have)
      if self.strict:
        assert need == have, (self.consec, self.length, self.prefix, have)
    start = self.index * self.length
    count = self.length + self.prefix
    first = self.current['is_first']
    if torch.is_tensor(first):
      # Device batches are always materialised contiguously (one kernel for all
      # keys); `contiguous` only matters for numpy views.
      if all(torch.is_tensor(v) and v.is_cuda for v in self.current.values()):
        chunk = window_batch(self.current, start, count)
      else:
        chunk = {k: window(v, start, count) for k, v in self.current.items()}
      chunk['consec'] = torch.full(
          chunk['is_first'].shape, self.index, dtype=torch.int32,
          device=first.device)
    else:
      chunk = {k: v[:, start: start + count] for k, v in self.current.items()}
      chunk['consec'] = np.full(chunk['is_first'].shape, self.index, np.int32)
      if self.contiguous:
        chunk = {k: np.ascontiguousarray(v) for k, v in chunk.items()}
    self.index += 1
    return chunk

  def _fused_source(self):
    """(replay, batch, mode) if the source is `Stateless(replay.sample, batch,
    mode)` over this package's Replay with matching lengths: then sampling and
    windowing run as one gather (`Replay.sample_windows`)."""
    if self._fused is not False:
      return self._fused
    self._fused = None
    fn = getattr(self.source, 'nextfn', None)
    target = getattr(getattr(fn, 'func', None), '__self__', None)
    from . import replay as replaylib
    if (isinstance(target, replaylib.Replay)
        and getattr(fn.func, '__func__', None) is replaylib.Replay.sample
        and not fn.keywords and 1 <= len(fn.args) <= 2
        and 'is_first' in (target._keyid or {'is_first': 0})
        and target.length == self.consec * self.length + self.prefix):
      mode = fn.args[1] if len(fn.args) > 1 else 'train'
      self._fused = (target, fn.args[0], mode)
    return self._fused

  def save(self):
    return {'source': self.source.save(), 'index': self.index}

  def load(self, data):
    self.source.load(data['source'])
    self.index = data['index']


class Prefetch(base.Stream):
  """`amount`-deep pipeline: a daemon thread pulls from the source and applies
  `transform` ahead of the consumer (streams.py:32-86)."""

  def __init__(self, source, transform=None, amount=1):
    self.source = iter(source) if hasattr(source, '__iter__') else source()
    self.transform = transform or (lambda x: x)
    self.state = self._getstate()
    self.requests = threading.Semaphore(amount)
    self.amount = amount
    self.queue = queue.Queue()
    self.worker = threading.Thread(target=self._worker, daemon=True)
    self.started = False

  def __iter__(self):
    assert not self.started
    self.worker.start()
    self.started = True
    return self

  def __next__(self):
    assert self.started
    result = self.queue.get()
    self.requests.release()
    if isinstance(result, BaseException):
      raise RuntimeError(str(result)) from result
    data, self.state = result
    return data

  def save(self):
    return self.state

  def load(self, state):
    if self.started:
      for _ in range(self.amount):
        self.queue.get()
    self.source.load(state)
    if self.started:
      self.requests.release(self.amount)

  def _worker(self):
    try:
      while True:
        self.requests.acquire()
        data = self.transform(next(self.source))
        self.queue.put((data, self._getstate()))
    except BaseException as e:
      self.queue.put(e)

  def _getstate(self):
    return self.source.save() if hasattr(self.source, 'save') else None
