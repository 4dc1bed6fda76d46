"""Index samplers with the reference's selector protocol
(`__call__`, `__len__`, `__setitem__(key, stepids)`, `__delitem__(key)`,
optional `prioritize(stepids, priorities)`; embodied/core/selectors.py).

The state lives in the host index core of libembodied_hip.so (C++; exact numpy
PCG64 streams, float64 tree sums), these classes are thin handles.  Item keys
are integers (Replay's item counter).
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import api


def _stepid_bytes(stepids):
  """(L,20) uint8 array, list of 20-byte strings or None -> (buffer, n)."""
  if stepids is None:
    return None, 0
  if hasattr(stepids, 'detach'):
    stepids = stepids.detach().cpu().numpy()
  if len(stepids) and isinstance(stepids[0], (bytes, bytearray)):
    flat = np.frombuffer(b''.join(stepids), np.uint8)
  else:
    flat = np.ascontiguousarray(stepids, np.uint8).reshape(-1)
  if flat.size % _lib.STEPID_BYTES:
    raise ValueError('step ids must be 20 bytes each')
  return flat, flat.size // _lib.STEPID_BYTES


class _Native:
  """Owns one emb_selector_t."""

  _handle = None

  def _adopt(self, handle):
    self._handle = handle

  def __del__(self):
    if self._handle is not None and api is not None:
      api.raw.emb_selector_destroy(self._handle)
      self._handle = None

  def __len__(self):
    n = C.c_int64()
    api.emb_selector_len(self._handle, C.byref(n))
    return n.value

  def __call__(self):
    key = C.c_int64()
    api.emb_selector_sample(self._handle, C.byref(key))
    return key.value

  def __setitem__(self, key, stepids):
    buf, n = _stepid_bytes(stepids)
    api.emb_selector_insert(self._handle, int(key), _lib.ptr(buf), n)

  def __delitem__(self, key):
    api.emb_selector_remove(self._handle, int(key))


class Fifo(_Native):
  """Always the oldest key (selectors.py:7-26)."""

  def __init__(self):
    handle = C.c_void_p()
    api.emb_selector_create_fifo(C.byref(handle))
    self._adopt(handle)


class Uniform(_Native):
  """Uniform over live keys, numpy-exact draws (selectors.py:29-57)."""

  def __init__(self, seed=0):
    handle = C.c_void_p()
    api.emb_selector_create_uniform(int(seed), C.byref(handle))
    self._adopt(handle)


class Prioritized(_Native):
  """Priority-proportional over per-step priorities (selectors.py:128-197)."""

  def __init__(self, exponent=1.0, initial=1.0, zero_on_sample=False,
               maxfrac=0.0, branching=16, seed=0):
    assert 0 <= maxfrac <= 1, maxfrac
    handle = C.c_void_p()
    api.emb_selector_create_prioritized(
        float(exponent), float(initial), int(bool(zero_on_sample)),
        float(maxfrac), int(branching), int(seed), C.byref(handle))
    self._adopt(handle)

  def __setitem__(self, key, stepids):
    if stepids is None:
      raise ValueError('Prioritized needs the step ids of every item')
    super().__setitem__(key, stepids)

  def prioritize(self, stepids, priorities):
    buf, n = _stepid_bytes(stepids)
    if hasattr(priorities, 'detach'):
      priorities = priorities.detach().cpu().numpy()
    prios = np.ascontiguousarray(priorities, np.float64).reshape(-1)
    if len(prios) != n:
      raise ValueError((len(prios), n))
    api.emb_selector_prioritize(self._handle, _lib.ptr(buf), _lib.ptr(prios), n)


class Mixture(_Native):
  """Chooses a member by fraction, then delegates (selectors.py:200-228).
  Unlike the reference it has `__len__`, without which Replay.sample cannot use
  it (replay.py:123)."""

  def __init__(self, selectors, fractions, seed=0):
    assert set(selectors.keys()) == set(fractions.keys())
    assert sum(fractions.values()) == 1, fractions
    names = sorted(k for k in selectors if fractions[k])
    self.selectors = [selectors[k] for k in names]
    self.fractions = np.array([fractions[k] for k in names], np.float32)
    members = [_as_native(s) for s in self.selectors]
    self._members = members                      # keep callback shims alive
    handles = (C.c_void_p * len(members))(*[m._handle for m in members])
    handle = C.c_void_p()
    api.emb_selector_create_mixture(
        handles, _lib.ptr(self.fractions), len(members), int(seed),
        C.byref(handle))
    self._adopt(handle)

  def prioritize(self, stepids, priorities):
    for member in self._members:
      if hasattr(member, 'prioritize'):
        member.prioritize(stepids, priorities)


class Foreign(_Native):
  """Adapts ANY object implementing the selector protocol so the native replay
  index can drive it through C callbacks."""

  def __init__(self, target):
    self.target = target
    self.error = None

    def guard(fn, default=None):
      def run(*args):
        try:
          return fn(*args)
        except BaseException as e:  # cannot unwind through C
          self.error = self.error or e
          return default
      return run

    def insert(_, key, stepids, n):
      ids = np.ctypeslib.as_array(stepids, (n, _lib.STEPID_BYTES)).copy() if n else None
      target[key] = ids

    def remove(_, key):
      del target[key]

    def prioritize(_, stepids, prios, n):
      ids = np.ctypeslib.as_array(stepids, (n, _lib.STEPID_BYTES)).copy()
      target.prioritize(ids, np.ctypeslib.as_array(prios, (n,)).copy())

    self._fns = (
        _lib.SAMPLE_FN(guard(lambda _: int(target()), 0)),
        _lib.SIZE_FN(guard(lambda _: len(target), 0)),
        _lib.INSERT_FN(guard(insert)),
        _lib.REMOVE_FN(guard(remove)),
        _lib.PRIORITIZE_FN(guard(prioritize)) if hasattr(target, 'prioritize')
        else _lib.PRIORITIZE_FN(),
    )
    self._table = _lib.SelectorCallbacks(None, *self._fns)
    handle = C.c_void_p()
    api.emb_selector_create_callback(C.byref(self._table), C.byref(handle))
    self._adopt(handle)
    if hasattr(target, 'prioritize'):
      self.prioritize = target.prioritize

  def reraise(self):
    if self.error is not None:
      error, self.error = self.error, None
      raise error


def _as_native(selector):
  return selector if isinstance(selector, _Native) else Foreign(selector)


class SampleTree:
  """b-ary sum tree (selectors.py:231-306)."""

  def __init__(self, branching=16, seed=0):
    assert 2 <= branching
    self.branching = branching
    self._handle = C.c_void_p()
    api.emb_tree_create(int(branching), int(seed), C.byref(self._handle))

  def __del__(self):
    if getattr(self, '_handle', None) is not None and api is not None:
      api.raw.emb_tree_destroy(self._handle)
      self._handle = None

  def __len__(self):
    n = C.c_int64()
    api.emb_tree_len(self._handle, C.byref(n))
    return n.value

  def insert(self, key, uprob):
    api.emb_tree_insert(self._handle, int(key), float(uprob))

  def remove(self, key):
    api.emb_tree_remove(self._handle, int(key))

  def update(self, key, uprob):
    api.emb_tree_update(self._handle, int(key), float(uprob))

  def sample(self):
    key = C.c_int64()
    api.emb_tree_sample(self._handle, C.byref(key))
    return key.value

  @property
  def total(self):
    out = C.c_double()
    api.emb_tree_root_sum(self._handle, C.byref(out))
    return out.value

  def shape(self):
    """(leaf depths, node count incl. leaves) — what the reference's tree
    tests inspect (tests/test_sampletree.py:18-58)."""
    n = len(self)
    depths = np.zeros(max(n, 1), np.int64)
    leaves, nodes = C.c_int64(), C.c_int64()
    api.emb_tree_shape(
        self._handle, len(depths), _lib.ptr(depths), C.byref(leaves),
        C.byref(nodes))
    return depths[:leaves.value], nodes.value


class Recency(_Native):
  """Age-biased sampling: the item inserted `age` inserts ago is drawn with
  probability proportional to `uprobs[age]` (reference: selectors.py:60-125).

  The reference implementation cannot draw as written (`_sample` reads an
  unbound `segment`, selectors.py:98-105); with that one token repaired
  (`len(p)`) its draws are the ones this class makes, bit for bit (golden
  `sel_recency`, tests/adapters.py builds the repaired class from the reference's
  own source).  The b-ary table of normalised block masses is built here with
  the reference's numpy arithmetic (`_build`, :107-125) and handed to the host
  index core, which draws one `choice` per level (csrc/selectors.h Recency) --
  native like the other members of a Mixture, no callbacks.
  """

  def __init__(self, uprobs, seed=0, bfactor=16):
    uprobs = np.asarray(uprobs, np.float64)
    assert uprobs[0] >= uprobs[-1], uprobs
    assert np.isfinite(uprobs).all() and (uprobs >= 0).all(), uprobs
    self.uprobs = uprobs
    self.bfactor = int(bfactor)
    self.levels = self._build(uprobs, self.bfactor)
    table = np.ascontiguousarray(np.concatenate([level.reshape(-1) for level in self.levels]), np.float64)
    handle = C.c_void_p()
    api.emb_selector_create_recency(
        _lib.ptr(table), table.size, len(self.levels), self.bfactor, len(uprobs), int(seed),
        C.byref(handle))
    self._adopt(handle)

  @staticmethod
  def _build(uprobs, bfactor):
    depth = max(1, int(np.ceil(np.log(len(uprobs)) / np.log(bfactor))))
    padded = np.zeros(bfactor ** depth)
    padded[:len(uprobs)] = uprobs
    levels = []
    masses = padded
    for _ in range(depth):
      groups = masses.reshape(-1, bfactor)
      totals = groups.sum(-1, keepdims=True)
      with np.errstate(divide='ignore', invalid='ignore'):
        probs = np.where(totals > 0, groups / totals, 1.0 / bfactor)
      levels.insert(0, probs)
      masses = totals[:, 0]
    return levels
