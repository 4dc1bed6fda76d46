"""`Space`: dtype/shape/bounds of one observation or action entry.

The reference takes this type from the un-vendored `elements` package
(`elements.Space`; used at embodied/envs/dummy.py:18-36, core/random.py:22,
core/wrappers.py:103,263).  Only the surface the hot path reads is provided:
`.dtype .shape .low .high .discrete .sample() __contains__`.
"""
import numpy as np


class Space:

  def __init__(self, dtype, shape=(), low=None, high=None):
    self._dtype = np.dtype(dtype)
    self._shape = (shape,) if isinstance(shape, int) else tuple(shape)
    self._discrete = (
        np.issubdtype(self._dtype, np.integer) or self._dtype == bool)
    self._low = self._bound(low, -1)
    self._high = self._bound(high, +1)
    self._rng = np.random.default_rng()

  def _bound(self, value, sign):
    # Bounds given by the caller are kept as given (only broadcast): wrappers
    # compute with them, and e.g. NormalizeAction's output dtype follows the
    # bounds' dtype.  Defaults: +-inf (float64) for floats, the dtype's own
    # range for integers, False/True for bool.
    if value is not None:
      return np.broadcast_to(np.asarray(value), self._shape)
    if np.issubdtype(self._dtype, np.floating):
      return np.broadcast_to(np.asarray(sign * np.inf), self._shape)
    if np.issubdtype(self._dtype, np.integer):
      info = np.iinfo(self._dtype)
      return np.broadcast_to(
          np.asarray(info.min if sign < 0 else info.max, self._dtype), self._shape)
    return np.broadcast_to(np.asarray(sign > 0), self._shape)

  @property
  def dtype(self):
    return self._dtype

  @property
  def shape(self):
    return self._shape

  @property
  def low(self):
    return self._low

  @property
  def high(self):
    return self._high

  @property
  def discrete(self):
    return self._discrete

  @property
  def nbytes(self):
    return int(np.prod(self._shape, dtype=np.int64)) * self._dtype.itemsize

  def __repr__(self):
    return (f'Space({self._dtype.name}, shape={self._shape}, '
            f'low={self._low.min() if self._low.size else None}, '
            f'high={self._high.max() if self._high.size else None})')

  def __contains__(self, value):
    value = np.asarray(value)
    if value.shape != self._shape:
      return False
    if not np.can_cast(value.dtype, self._dtype, 'same_kind'):
      return False
    if self._dtype == bool:
      return True
    return bool((value >= self._low).all() and (value <= self._high).all())

  def sample(self):
    if self._dtype == bool:
      return self._rng.random(self._shape) > 0.5
    if self._discrete:
      # high is exclusive for integer spaces, as numpy's randint convention.
      return self._rng.integers(
          self._low, self._high, self._shape, dtype=np.int64).astype(self._dtype)
    info = np.finfo(self._dtype)
    low = np.maximum(self._low, info.min / 2)
    high = np.minimum(self._high, info.max / 2)
    return self._rng.uniform(low, high, self._shape).astype(self._dtype)
