"""Random envs and policies through the Driver (host mode here, device mode
under -m gpu) against the oracle's Driver, which the golden `driver_script`
pins to the reference: transition order, masking of every action key on the
steps that end an episode, reset flags, log/ keys, episode counting."""
import numpy as np
import pytest

import embodied_amd as emb
from oracle import np_oracle


class RandomEnv:
  """Episode ends, terminal flags and observations from a private generator."""

  def __init__(self, seed, vec=3):
    self.gen = np.random.default_rng(seed)
    self.vec = vec
    self.t = 0

  @property
  def obs_space(self):
    return {'vec': emb.Space(np.float32, (self.vec,)), 'count': emb.Space(np.int32),
            'reward': emb.Space(np.float32), 'is_first': emb.Space(bool),
            'is_last': emb.Space(bool), 'is_terminal': emb.Space(bool), 'log/extra': emb.Space(np.float32)}

  @property
  def act_space(self):
    return {'move': emb.Space(np.float32, (2,), -1, 1), 'pick': emb.Space(np.int32, (), 0, 5),
            'reset': emb.Space(bool)}

  def step(self, action):
    first = bool(action['reset'])
    self.t = 0 if first else self.t + 1
    last = (not first) and self.gen.random() < 0.25
    return {
        'vec': (self.gen.standard_normal(self.vec) + np.asarray(action['move']).sum()).astype(np.float32),
        'count': np.int32(self.t), 'reward': np.float32(np.asarray(action['pick']) * 0.5),
        'is_first': first, 'is_last': bool(last), 'is_terminal': bool(last and self.gen.random() < 0.5),
        'log/extra': np.float32(self.t * 2)}

  def close(self):
    pass


def _run(make_driver, to_np, seed, n_env, steps):
  gen = np.random.default_rng(seed)
  driver = make_driver([RandomEnv(1000 * seed + i) for i in range(n_env)])
  log = []
  driver.on_step(lambda tran, worker, **kw: log.append(
      (worker, {k: np.array(to_np(v)) for k, v in tran.items()})))

  def policy(carry, obs, **kw):
    n = len(to_np(obs['is_first']))
    assert 'log/extra' not in obs
    move = gen.standard_normal((n, 2)).astype(np.float32)
    move[0, 0] = -abs(move[0, 0])                       # a negative entry: masks to -0.0
    return carry + 1, {'move': move, 'pick': gen.integers(0, 5, n).astype(np.int32)}, {
        'value': gen.standard_normal(n).astype(np.float32)}

  driver.reset(lambda n: 0)
  driver(policy, steps=steps)
  driver(policy, episodes=3)
  return log, driver.carry


@pytest.mark.parametrize('seed', range(5))
def test_host_driver_equals_oracle_driver(seed):
  n_env = 1 + seed % 4
  ours, carry_a = _run(lambda envs: emb.Driver([(lambda e=e: e) for e in envs], parallel=False),
                       np.asarray, seed, n_env, 40)
  want, carry_b = _run(np_oracle.Driver, np.asarray, seed, n_env, 40)
  assert carry_a == carry_b and len(ours) == len(want)
  for (wa, ta), (wb, tb) in zip(ours, want):
    assert wa == wb and set(ta) == set(tb)
    for k in tb:
      assert ta[k].dtype == tb[k].dtype and ta[k].tobytes() == tb[k].tobytes(), k


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(3))
def test_device_driver_equals_oracle_driver(seed):
  import torch
  n_env = 2 + seed
  to_np = lambda v: v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
  ours, carry_a = _run(lambda envs: emb.Driver([(lambda e=e: e) for e in envs], parallel=False, device='cuda'),
                       to_np, seed, n_env, 40)
  want, carry_b = _run(np_oracle.Driver, np.asarray, seed, n_env, 40)
  assert carry_a == carry_b and len(ours) == len(want)
  for (wa, ta), (wb, tb) in zip(ours, want):
    assert wa == wb and set(ta) == set(tb)
    for k in tb:
      assert ta[k].dtype == tb[k].dtype and ta[k].tobytes() == tb[k].tobytes(), k
