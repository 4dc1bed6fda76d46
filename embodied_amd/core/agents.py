"""Agents that need no learning machinery.

`RandomAgent` is the plumbing agent of the reference (embodied/core/random.py):
uniform actions from the action spaces, empty carries, no-op train/report.  It
lets Driver / Replay / run loops be exercised without a model.
"""
import numpy as np

from . import base


class RandomAgent(base.Agent):

  def __init__(self, obs_space, act_space, config=None):
    self.obs_space = obs_space
    self.act_space = act_space
    self._sampled = [k for k in act_space if k != 'reset']

  # Carries are empty tuples for all three roles.
  def _no_carry(self, batch_size):
    return ()

  init_policy = init_train = init_report = _no_carry

  def policy(self, carry, obs, mode='train'):
    envs = len(obs['is_first'])
    act = {}
    for name in self._sampled:
      space = self.act_space[name]
      draws = np.empty((envs, *space.shape), space.dtype)
      for row in range(envs):
        draws[row] = space.sample()
      act[name] = draws
    return carry, act, {}

  def train(self, carry, data):
    return carry, {}, {}

  def report(self, carry, data):
    return carry, {}

  def stream(self, st):
    return st

  def save(self):
    return None

  def load(self, data=None):
    return None
