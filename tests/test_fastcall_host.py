"""The CPython call shim's `columns` helper (Replay.add_batch's per-key checks in
C) on CPU tensors: `get_device()` is -1 there, so device -1 selects them."""
import ctypes as C

import numpy as np
import pytest
import torch

from embodied_amd import _lib


@pytest.fixture(scope='module')
def columns():
  if _lib.fast.columns is None:
    pytest.skip('call shim not built (ctypes binding only)')
  return _lib.fast.columns


def test_columns_takes_ready_tensors_and_reports_the_rest(columns):
  n = 4
  steps = {
      'image': torch.zeros((n, 3, 2), dtype=torch.uint8),
      'reward': torch.zeros(n, dtype=torch.float32),
      'log/x': torch.zeros(n),                               # not stored
      'wrong_dtype': torch.zeros(n, dtype=torch.float64),
      'wrong_shape': torch.zeros((n, 2), dtype=torch.float32),
      'strided': torch.zeros((n, 8), dtype=torch.float32)[:, ::2],
      'array': np.zeros(n, np.float32),
      'flag': torch.zeros(n, dtype=torch.bool),
  }
  plan = (
      (2, torch.uint8, (n, 3, 2), 'image'), (0, torch.float32, (n,), 'reward'),
      (-1, None, None, 'log/x'), (1, torch.float32, (n,), 'wrong_dtype'),
      (3, torch.float32, (n,), 'wrong_shape'), (4, torch.float32, (n, 4), 'strided'),
      (5, torch.float32, (n,), 'array'), (6, torch.bool, (n,), 'flag'))
  out = (C.c_void_p * 7)()
  slow = columns(steps, plan, out, torch.Tensor, -1)
  assert slow == [3, 4, 5, 6]
  assert out[2] == steps['image'].data_ptr() and out[0] == steps['reward'].data_ptr()
  assert out[6] == steps['flag'].data_ptr()
  assert out[1] is None and out[3] is None and out[4] is None and out[5] is None
  # another device index: nothing is ready
  out2 = (C.c_void_p * 7)()
  assert columns(steps, plan, out2, torch.Tensor, 0) == [0, 1, 3, 4, 5, 6, 7]
  assert all(out2[i] is None for i in range(7))
  # everything ready -> None
  ready = {'a': steps['reward'], 'b': steps['flag']}
  assert columns(ready, ((0, torch.float32, (n,), 'a'), (1, torch.bool, (n,), 'b')),
                 (C.c_void_p * 2)(), torch.Tensor, -1) is None


def test_columns_rejects_bad_calls(columns):
  t = torch.zeros(3)
  with pytest.raises(ValueError):
    columns({'a': t, 'b': t}, ((0, torch.float32, (3,), 'a'),), (C.c_void_p * 2)(), torch.Tensor, -1)
  with pytest.raises(IndexError):
    columns({'a': t}, ((5, torch.float32, (3,), 'a'),), (C.c_void_p * 2)(), torch.Tensor, -1)
  with pytest.raises(TypeError):
    columns([t], ((0, torch.float32, (3,), 'a'),), (C.c_void_p * 2)(), torch.Tensor, -1)
  with pytest.raises((TypeError, BufferError)):
    columns({'a': t}, ((0, torch.float32, (3,), 'a'),), b'readonly', torch.Tensor, -1)
  # subclasses take the slow path (type(value) is Tensor in the Python loop)
  class Sub(torch.Tensor):
    pass
  sub = torch.zeros(3).as_subclass(Sub)
  assert columns({'a': sub}, ((0, torch.float32, (3,), 'a'),), (C.c_void_p * 1)(), torch.Tensor, -1) == [0]
