"""Return scans on device (float32): GAE (ppo/agent.py:188-201), lambda-return
(dreamerv3/agent.py:482-490) and the Director critic target
(director/agent.py:430-445), each one kernel launch.

Inputs are torch CUDA tensors; bool flags may be torch.bool or uint8.
"""
import os
import sys

import numpy as np
import torch

from . import _lib
from ._lib import api, fast


def _stream(t):
  return _lib.raw_stream(t.device)


_ROUNDED = {}


def _round32(x):
  """x rounded to float32 (what the float32 reference computes with), as a
  Python float; the handful of hyper-parameter values is remembered."""
  try:
    return _ROUNDED[x]
  except KeyError:
    if len(_ROUNDED) > 1024:
      _ROUNDED.clear()
    value = _ROUNDED[x] = float(np.float32(x))
    return value


# A tensor OBJECT that passed `_f32` / `_flag` for a device keeps a mark (the
# replay hands out the same output tensors again once nobody holds them, a
# critic's value buffer is often one tensor): the next call recognises it with
# one attribute read instead of four calls into torch.  dtype, device and
# strides of a tensor do not change behind its back (short of resize_ / set_).
_F32_OK, _FLAG_OK = {}, {}


def _mark(table, device):
  mark = table.get(device)
  if mark is None:
    mark = table[device] = (device,)
  return mark


def _f32(x, device):
  if type(x) is torch.Tensor:
    mark = _mark(_F32_OK, device)
    if getattr(x, '_emb_f32', None) is mark:
      return x
    if x.dtype == torch.float32 and x.device == device and x.is_contiguous():
      x._emb_f32 = mark
      return x
  if torch.is_tensor(x) and x.dtype == torch.float32 and x.device == device and x.is_contiguous():
    return x
  if not torch.is_tensor(x):
    x = torch.as_tensor(np.asarray(x))
  return x.to(device=device, dtype=torch.float32).contiguous()


def _flag(x, device):
  """bool / uint8 flags as a contiguous 1-byte tensor on `device` (the kernels
  read the bytes; torch.bool is one byte per element)."""
  if type(x) is torch.Tensor:
    mark = _mark(_FLAG_OK, device)
    if getattr(x, '_emb_flag', None) is mark:
      return x
    if (x.device == device and x.is_contiguous()
        and (x.dtype == torch.bool or x.dtype == torch.uint8)):
      x._emb_flag = mark
      return x
  if (torch.is_tensor(x) and x.device == device and x.is_contiguous()
      and (x.dtype == torch.bool or x.dtype == torch.uint8)):
    return x
  if not torch.is_tensor(x):
    x = torch.as_tensor(np.asarray(x))
  if x.dtype != torch.bool and x.dtype != torch.uint8:
    x = x != 0
  return x.to(device).contiguous()


def _device(*xs):
  for x in xs:
    if torch.is_tensor(x) and x.is_cuda:
      return x.device
  raise RuntimeError(
      'embodied_amd.scans run as HIP kernels: pass CUDA tensors (no CPU fallback)')


def _pair(B, n, dev):
  """Two fresh (B, n) float32 results out of one (2, B, n) allocation (`out=`
  is the form without an allocation)."""
  return _lib.empty((2, B, n), torch.float32, dev).unbind(0)


def gae(rew, val, last, term, hor=200, lam=0.8, out=None):
  """adv_t = delta_t + live_t*cont_t*adv_{t+1}; tar = adv + val[:, :-1].
  rew, val (B,T) f32; last, term (B,T) bool -> adv, tar (B,T-1).
  `out=(adv, tar)`: write into the caller's contiguous float32 (B,T-1) tensors."""
  dev = _device(rew, val, last, term)
  rew, val = _f32(rew, dev), _f32(val, dev)
  last, term = _flag(last, dev), _flag(term, dev)
  B, T = rew.shape
  assert val.shape == last.shape == term.shape == (B, T)
  if out is not None:
    adv, tar = out
    for result in (adv, tar):
      # (a tensor object that passed for this shape and device keeps a mark, like
      # the inputs: an agent hands the same few result tensors in again)
      if result.__dict__.get('_emb_gae_out') == (B, T, dev):
        continue
      if (result.dtype != torch.float32 or tuple(result.shape) != (B, T - 1) or result.device != dev
          or not result.is_contiguous()):
        raise ValueError(f'gae(out=): needs contiguous float32 {(B, T - 1)} tensors on {dev}')
      result._emb_gae_out = (B, T, dev)
  elif B * T <= 1 << 20:
    adv, tar = _pair(B, T - 1, dev)                         # one allocation, two views
  else:       # bandwidth-bound sizes: two write streams a power-of-two-ish distance apart
    adv = _lib.empty((B, T - 1), torch.float32, dev)        # collide on HBM channels (-19 %)
    tar = _lib.empty((B, T - 1), torch.float32, dev)
  if B == 0 or T < 2:
    return adv, tar                   # nothing to scan: (B, 0) results
  fast.emb_scan_gae(
      rew.data_ptr(), val.data_ptr(), last.data_ptr(), term.data_ptr(), B, T,
      _round32(1 - 1 / hor), _round32(lam), adv.data_ptr(),
      tar.data_ptr(), _stream(rew))
  return adv, tar


def lambda_return(last, term, rew, val, boot, disc, lam):
  """ret_t = interm_t + live_t*cont_t*ret_{t+1}, seeded with boot[:, -1].
  All (B,T) -> (B,T-1).  `val` is only shape-checked, as in the reference."""
  dev = _device(rew, boot, last, term)
  rew, boot = _f32(rew, dev), _f32(boot, dev)
  last, term = _flag(last, dev), _flag(term, dev)
  B, T = rew.shape
  assert boot.shape == last.shape == term.shape == (B, T)
  assert val is None or tuple(val.shape) == (B, T)
  ret = _lib.empty((B, T - 1), torch.float32, dev)
  if B == 0 or T < 2:
    return ret
  fast.emb_scan_lambda(
      last.data_ptr(), term.data_ptr(), rew.data_ptr(), boot.data_ptr(), B, T,
      _round32(disc), _round32(lam), ret.data_ptr(),
      _stream(rew))
  return ret


_MULTI = {}


def lambda_returns(problems, out=None):
  """Several `lambda_return` problems of one train step in ONE launch:
  `problems` = [(last, term, rew, val, boot, disc, lam), ...] (the arguments of
  `lambda_return`), result = [ret, ...].  DreamerV3 computes the replay returns
  (B, T) and the imagined returns (B*K, H+1) in the same train step
  (dreamerv3/agent.py:401-405, 464-466); at those sizes each scan is launch
  latency, so one launch costs half of two.  `out=[ret, ...]`: the caller's
  contiguous float32 (B, T-1) tensors."""
  import ctypes as C
  prepared, shapes = [], []
  for last, term, rew, val, boot, disc, lam in problems:
    dev = _device(rew, boot, last, term)
    rew, boot = _f32(rew, dev), _f32(boot, dev)
    last, term = _flag(last, dev), _flag(term, dev)
    B, T = rew.shape
    assert boot.shape == last.shape == term.shape == (B, T)
    assert val is None or tuple(val.shape) == (B, T)
    prepared.append((last, term, rew, boot, dev))
    shapes.append((B, T, _round32(disc), _round32(lam)))
  key = tuple(shapes)
  table = _MULTI.get(key)
  if table is None:
    if len(_MULTI) > 64:
      _MULTI.clear()
    table = _MULTI[key] = (_lib.LambdaProblem * len(shapes))()
    for entry, (B, T, disc, lam) in zip(table, shapes):
      entry.B, entry.T, entry.disc, entry.lam = B, T, disc, lam
  rets = []
  for i, ((last, term, rew, boot, dev), (B, T, _, _)) in enumerate(zip(prepared, shapes)):
    if out is not None:
      ret = out[i]
      if (ret.dtype != torch.float32 or tuple(ret.shape) != (B, T - 1) or ret.device != dev
          or not ret.is_contiguous()):
        raise ValueError(f'lambda_returns(out=): needs contiguous float32 {(B, T - 1)} tensors on {dev}')
    else:
      ret = _lib.empty((B, T - 1), torch.float32, dev)
    entry = table[i]
    entry.last, entry.term, entry.rew = last.data_ptr(), term.data_ptr(), rew.data_ptr()
    entry.boot, entry.ret = boot.data_ptr(), ret.data_ptr()
    rets.append(ret)
  if prepared:
    fast.emb_scan_lambda_multi(len(prepared), table, _stream(prepared[0][2]))
  return rets


def director_score(rew, cont, value, horizon=333, lam=0.95):
  """Time-major: rew (T-1,B), cont, value (T,B) -> ret (T-1,B)."""
  dev = _device(rew, cont, value)
  rew, cont, value = _f32(rew, dev), _f32(cont, dev), _f32(value, dev)
  T, B = value.shape
  assert cont.shape == (T, B) and rew.shape == (T - 1, B)
  ret = _lib.empty((T - 1, B), torch.float32, dev)
  if B == 0 or T < 2:
    return ret
  api.emb_scan_director(
      rew.data_ptr(), cont.data_ptr(), value.data_ptr(), T, B,
      _round32(1 - 1 / horizon), _round32(lam),
      ret.data_ptr(), _stream(rew))
  return ret


def split_traj(x, k, is_reward=False):
  """Director worker windows (director/hierarchy.py:224-238): time-major
  (T,B,...) -> (k, (T/k)*B, ...) views/reshapes; reward keys shift by one."""
  if is_reward:
    x = torch.cat([0 * x[:1], x], 0)
  x = x.reshape((x.shape[0] // k, k) + tuple(x.shape[1:]))
  x = x.transpose(0, 1)
  x = x.reshape((x.shape[0], -1) + tuple(x.shape[3:]))
  return x[1:] if is_reward else x


def abstract_traj(x, cont, k, kind='first'):
  """Director manager steps (director/hierarchy.py:240-256), time-major.
  kind 'reward': x (T-1,B) -> cumprod(cont)-weighted window means (T/k-1,B);
  'cont': x = cont (T,B) -> window products (T/k,B) — both one kernel
  (`emb_abstract_traj`); anything else: first step of every window (a view)."""
  if kind in ('reward', 'cont'):
    if not (torch.is_tensor(cont) and cont.is_cuda and cont.dim() == 2):
      raise RuntimeError(
          'abstract_traj reward/cont windows run as a HIP kernel: `cont` must be a (T, B) CUDA '
          'tensor (no CPU fallback)')
    dev = cont.device
    c = _f32(cont, dev)
    T, B = c.shape
    assert T % k == 0, (T, k)
    if kind == 'reward':
      r = _f32(x, dev)
      assert r.shape == (T - 1, B), (r.shape, c.shape)
      out = torch.empty((T // k - 1, B), dtype=torch.float32, device=dev)
      api.emb_abstract_traj(r.data_ptr(), c.data_ptr(), T, B, k, out.data_ptr(), None, _stream(c))
    else:
      out = torch.empty((T // k, B), dtype=torch.float32, device=dev)
      api.emb_abstract_traj(None, c.data_ptr(), T, B, k, None, out.data_ptr(), _stream(c))
    return out
  return x.reshape((x.shape[0] // k, k) + tuple(x.shape[1:]))[:, 0]   # first step of each window
