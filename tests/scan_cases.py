"""Seeded inputs of the return-scan fixtures.

Shared by `oracle/gen_scan_golden.py` (which feeds them to the reference's own
source) and by the tests (which regenerate them and check the digest stored in
the fixture, so a fixture can never be compared against other inputs).
"""
import hashlib

import numpy as np

f32 = np.float32

SEEDS = (0, 1, 2, 3)
SHAPES_BT = ((16, 64), (1024, 16), (16, 1024))       # batch-major (B, T)
SHAPES_TB = ((64, 16), (16, 1024), (1024, 16))       # Director: time-major (T, B)
GAE_PARAMS = dict(hor=200, lam=0.8)                    # ppo/configs.yaml:97
LAMBDA_PARAMS = (                                      # dreamerv3/agent.py:387-389
    dict(disc=1 - 1 / 333, lam=0.95),
    dict(disc=1.0, lam=0.95),
)
DIRECTOR_PARAMS = dict(horizon=333, lam=0.95)          # director/configs.yaml:125-126
SKILL_DURATION = 8                                     # director/configs.yaml:83
TRAJ_SHAPES_TB = ((16, 1024), (64, 16))                # (H+1, B*T) with H=15; a long one


def batch_major(seed, shape):
  """rew, val, boot ~ N(0,1) f32; last ~ Bernoulli(.02), term ~ Bernoulli(.01)."""
  rng = np.random.default_rng([seed, *shape])
  return dict(
      rew=rng.standard_normal(shape).astype(f32),
      val=rng.standard_normal(shape).astype(f32),
      boot=rng.standard_normal(shape).astype(f32),
      last=rng.random(shape) < 0.02,
      term=rng.random(shape) < 0.01)


def time_major(seed, shape):
  """Director critic inputs: rew (T-1,B), value (T,B) ~ N(0,1); cont (T,B) is
  1 except Bernoulli(.02) zeros (and a few fractional values, as the predicted
  continuation probability is)."""
  T, B = shape
  rng = np.random.default_rng([seed, T, B, 7])
  cont = np.where(rng.random(shape) < 0.02, 0.0, 1.0)
  cont = np.where(rng.random(shape) < 0.05, rng.random(shape), cont).astype(f32)
  return dict(
      rew=rng.standard_normal((T - 1, B)).astype(f32),
      value=rng.standard_normal(shape).astype(f32),
      cont=cont)


def trajectory(seed, shape, feat=3):
  """An imagined trajectory as Director's split_traj/abstract_traj see it."""
  T, B = shape
  data = time_major(seed, shape)
  rng = np.random.default_rng([seed, T, B, 11])
  return {
      'cont': data['cont'],
      'reward_extr': data['rew'],
      'reward_goal': rng.standard_normal((T - 1, B)).astype(f32),
      'action': rng.standard_normal((T, B, feat)).astype(f32),
      'weights': rng.random((T, B)).astype(f32),
  }


def digest(arrays):
  """sha256 over the named arrays' dtype, shape and bytes, as 32 uint8."""
  h = hashlib.sha256()
  for name in sorted(arrays):
    a = np.ascontiguousarray(arrays[name])
    h.update(name.encode() + str(a.dtype).encode() + str(a.shape).encode())
    h.update(a.tobytes())
  return np.frombuffer(h.digest(), np.uint8).copy()
