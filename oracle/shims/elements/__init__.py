"""Stand-in for the *used surface* of the un-vendored `elements` package.

TEST INFRASTRUCTURE ONLY.  This is NOT the reference and NOT the product: it
exists so that `oracle/gen_golden.py` can import `/root/reference/embodied/core`
unmodified in the build container (SURVEY.md Appendix A lists the symbols).
Anything whose value depends on this stand-in (the 16 UUID bytes inside
`stepid`, chunk file names, timer stats) is *not* pinned by the golden vectors.
"""
import contextlib
import itertools
import pathlib
import string
import threading
import time
import uuid as uuidlib

import numpy as np


class _Timer:

  def section(self, name):
    return _Section()

  def stats(self):
    return {'summary': ''}


class _Section(contextlib.ContextDecorator):

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    return False


timer = _Timer()


class RWLock:
  """Readers share, writers exclude (behavioural stand-in)."""

  def __init__(self):
    self._cond = threading.Condition()
    self._readers = 0
    self._writer = False

  @property
  @contextlib.contextmanager
  def reading(self):
    with self._cond:
      while self._writer:
        self._cond.wait()
      self._readers += 1
    try:
      yield
    finally:
      with self._cond:
        self._readers -= 1
        self._cond.notify_all()

  @property
  @contextlib.contextmanager
  def writing(self):
    with self._cond:
      while self._writer or self._readers:
        self._cond.wait()
      self._writer = True
    try:
      yield
    finally:
      with self._cond:
        self._writer = False
        self._cond.notify_all()


_ALPHABET = string.digits + string.ascii_letters


class UUID:

  _counter = None
  _lock = threading.Lock()

  @classmethod
  def reset(cls, *, debug):
    with cls._lock:
      cls._counter = itertools.count(1) if debug else None

  def __init__(self, value=None):
    if value is None:
      with self._lock:
        if self._counter is None:
          self.value = uuidlib.uuid4().int
        else:
          self.value = next(self._counter)
    elif isinstance(value, UUID):
      self.value = value.value
    elif isinstance(value, (int, np.integer)):
      self.value = int(value)
    elif isinstance(value, (bytes, bytearray)):
      self.value = int.from_bytes(bytes(value), 'big')
    elif isinstance(value, str):
      number = 0
      for char in value:
        number = number * 62 + _ALPHABET.index(char)
      self.value = number
    else:
      raise TypeError(type(value))

  def __int__(self):
    return self.value

  def __bytes__(self):
    return self.value.to_bytes(16, 'big')

  def __str__(self):
    number, chars = self.value, []
    while number:
      number, rem = divmod(number, 62)
      chars.append(_ALPHABET[rem])
    return ''.join(reversed(chars)).rjust(22, '0')

  def __repr__(self):
    return f'UUID({self.value})'

  def __hash__(self):
    return hash(self.value)

  def __eq__(self, other):
    return isinstance(other, UUID) and self.value == other.value

  def __lt__(self, other):
    return self.value < other.value


def timestamp(now=None, millis=False):
  now = time.time() if now is None else now
  base = time.strftime('%Y%m%dT%H%M%S', time.gmtime(now))
  if millis:
    return base + f'F{int((now % 1) * 1000):03d}'
  return base


class Path(type(pathlib.Path())):

  def mkdir(self, *args, **kwargs):
    kwargs.setdefault('parents', True)
    kwargs.setdefault('exist_ok', True)
    return super().mkdir(*args, **kwargs)

  def write(self, content, mode='w'):
    with open(self, mode) as f:
      f.write(content)

  def read(self, mode='r'):
    with open(self, mode) as f:
      return f.read()

  def read_bytes(self):
    return self.read('rb')


class _Tree:

  def map(self, fn, *trees, isleaf=None):
    first = trees[0]
    if isleaf and isleaf(first):
      return fn(*trees)
    if isinstance(first, dict):
      return {k: self.map(fn, *[t[k] for t in trees], isleaf=isleaf)
              for k in first}
    if isinstance(first, (list, tuple)):
      out = [self.map(fn, *xs, isleaf=isleaf) for xs in zip(*trees)]
      return type(first)(out) if not hasattr(first, '_fields') else type(
          first)(*out)
    return fn(*trees)


tree = _Tree()


class Space:

  def __init__(self, dtype, shape=(), low=None, high=None):
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    self._dtype = np.dtype(dtype)
    self._shape = shape
    if low is None:
      low = self._default(-1)
    if high is None:
      high = self._default(+1)
    self._low = np.broadcast_to(np.asarray(low), shape)
    self._high = np.broadcast_to(np.asarray(high), shape)
    self._discrete = (
        np.issubdtype(self._dtype, np.integer) or self._dtype == bool)
    self._random = np.random.RandomState()

  def _default(self, sign):
    if np.issubdtype(self._dtype, np.floating):
      return sign * np.inf
    if np.issubdtype(self._dtype, np.integer):
      info = np.iinfo(self._dtype)
      return info.min if sign < 0 else info.max
    return sign > 0

  dtype = property(lambda self: self._dtype)
  shape = property(lambda self: self._shape)
  low = property(lambda self: self._low)
  high = property(lambda self: self._high)
  discrete = property(lambda self: self._discrete)

  def __repr__(self):
    return f'Space({self._dtype.name}, shape={self._shape})'

  def __contains__(self, value):
    value = np.asarray(value)
    if value.shape != self._shape:
      return False
    if not np.can_cast(value.dtype, self._dtype, 'same_kind'):
      return False
    return bool((value >= self._low).all() and (value <= self._high).all())

  def sample(self):
    low, high = self._low, self._high
    if np.issubdtype(self._dtype, np.floating):
      info = np.finfo(self._dtype)
      low = np.maximum(info.min, low)
      high = np.minimum(info.max, high)
      return self._random.uniform(low, high, self._shape).astype(self._dtype)
    if self._dtype == bool:
      return self._random.uniform(0, 1, self._shape) > 0.5
    return self._random.randint(low, high, self._shape).astype(self._dtype)
