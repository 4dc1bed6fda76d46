"""Driver-side device ops: observation stack/transpose into the policy batch
and per-env carry rows by env id (embodied/core/driver.py:65,
embodied/jax/agent.py:173-181,230)."""
import numpy as np
import torch

from . import _lib
from ._lib import api, fast

_OUT = {torch.uint8: _lib.U8, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16,
        torch.float32: _lib.F32}


def _stream(t):
  return _lib.raw_stream(t.device)


def obs_stack(frames, env_ids=None, layout='channels_first', dtype=torch.uint8,
              scale=1.0, offset=0.0, out=None):
  """frames: uint8 (N_total, H, W, C) slab on the GPU.  Returns the policy batch
  (n, C, H, W) [channels_first] or (n, H, W, C) [same] as `dtype`, values
  `x * scale + offset` for float outputs; batch row j is env `env_ids[j]`.
  `out`: a contiguous tensor of that shape and dtype to write into (an agent
  that owns its input staging saves the allocation, ~2.5 us of host time)."""
  if env_ids is None and out is not None and type(frames) is torch.Tensor:
    # The same frame tensor into the same staging tensor with the same options as
    # before (a vector env's output ring, an agent's staging buffers): both
    # passed the checks below then, and a tensor's dtype, shape, device and
    # strides do not change behind its back -- see Replay._collect.
    memo = frames.__dict__.get('_emb_stack')
    if (memo is not None and memo[0] is out and memo[1] == layout and memo[2] is dtype
        and memo[3] == scale and memo[4] == offset):
      _, _, _, scale_f, offset_f, n, pixels, c, first = memo[:9]
      tag = frames.__dict__.get('_emb_offer')
      if tag is not None:
        replay = tag()
        if replay is not None and replay._early_insert(
            frames, pixels, c, first, dtype, scale_f, offset_f, out, memo):
          return out
      fast.emb_obs_stack(
          frames.data_ptr(), None, n, pixels, c,
          _lib.LAYOUT_CHANNELS_FIRST if first else _lib.LAYOUT_SAME, _OUT[dtype],
          scale_f, offset_f, out.data_ptr(), _stream(frames))
      return out
  if not (torch.is_tensor(frames) and frames.is_cuda and frames.dtype == torch.uint8):
    raise RuntimeError('obs_stack needs a uint8 CUDA tensor (no CPU fallback)')
  if not frames.is_contiguous():
    frames = frames.contiguous()
  total, h, w, c = frames.shape
  ids = None if env_ids is None else np.ascontiguousarray(env_ids, np.int32)
  n = total if ids is None else len(ids)
  first = layout == 'channels_first'
  shape = (n, c, h, w) if first else (n, h, w, c)
  if out is None:
    out = _lib.empty(shape, dtype, frames.device)
  elif not (out.shape == shape and out.dtype is dtype and out.is_contiguous()
            and out.get_device() == frames.get_device()):
    raise ValueError(f'obs_stack out= must be contiguous {shape} {dtype} on {frames.device}')
  if out.numel() == 0:
    return out                       # no envs / empty frames: nothing to launch
  if ids is None and type(frames) is torch.Tensor:
    frames._emb_stack = (out, layout, dtype, float(scale), float(offset), n, h * w, c, first)
  tag = getattr(frames, '_emb_offer', None)
  if tag is not None and ids is None:
    # A Driver offered this step's observations to its Replay (Replay.offer):
    # the launch that builds the policy batch also writes the observation keys
    # into the step's pool rows, every frame is read once.
    replay = tag()
    if replay is not None and replay._early_insert(
        frames, h * w, c, first, dtype, float(scale), float(offset), out,
        frames.__dict__.get('_emb_stack')):
      return out
  fast.emb_obs_stack(
      frames.data_ptr(), _lib.ptr(ids), n, h * w, c,
      _lib.LAYOUT_CHANNELS_FIRST if first else _lib.LAYOUT_SAME, _OUT[dtype],
      float(scale), float(offset), out.data_ptr(), _stream(frames))
  return out


def rows_gather(table, ids):
  """out[j] = table[ids[j]] (rows are whatever follows the first axis)."""
  table = table.contiguous()
  ids = np.ascontiguousarray(ids, np.int32)
  out = torch.empty((len(ids), *table.shape[1:]), dtype=table.dtype, device=table.device)
  rowbytes = table.element_size() * int(np.prod(table.shape[1:], dtype=np.int64))
  if out.numel() == 0:
    return out
  api.emb_rows_gather(table.data_ptr(), rowbytes, _lib.ptr(ids), len(ids),
                      out.data_ptr(), _stream(table))
  return out


def rows_scatter(table, ids, rows):
  """table[ids[j]] = rows[j], in place."""
  assert table.is_contiguous()
  ids = np.ascontiguousarray(ids, np.int32)
  rows = rows.to(table.device, table.dtype).contiguous()
  rowbytes = table.element_size() * int(np.prod(table.shape[1:], dtype=np.int64))
  if rows.numel() == 0:
    return table
  api.emb_rows_scatter(table.data_ptr(), rowbytes, _lib.ptr(ids), len(ids),
                       rows.data_ptr(), _stream(table))
  return table


class CuStream:
  """A HIP stream whose kernels may use `n_cus` compute units only (from unit
  `first_cu` on; the driver deals consecutive units round-robin over the XCDs),
  as a torch stream: `with torch.cuda.stream(cs.stream): ...`.  For a learner
  whose sample gather / scans / write-back run BESIDE the Driver's latency-bound
  step kernels instead of in front of them (`emb_stream_create_on_cus`,
  include/embodied_hip.h); the replay's movers size their grids by the share."""

  def __init__(self, n_cus, first_cu=0, device=None):
    import ctypes as C
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
      device = torch.device('cuda', torch.cuda.current_device())
    self.device, self.n_cus, self.first_cu = device, int(n_cus), int(first_cu)
    self._raw = C.c_void_p()
    with torch.cuda.device(device):
      api.emb_stream_create_on_cus(self.first_cu, self.n_cus, C.byref(self._raw))
    self.stream = torch.cuda.ExternalStream(self._raw.value, device=device)

  def close(self):
    raw, self._raw = self._raw, None
    if raw:
      self.stream.synchronize()
      api.emb_stream_destroy(raw)

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
