"""Fixture env with the behaviour of the reference's Dummy
(embodied/envs/dummy.py): fixed-length episodes, constant observations and a
`count` channel agents use to assert step continuity."""
import numpy as np

from ..core import base
from ..space import Space


class Dummy(base.Env):

  def __init__(self, task='disc', size=(64, 64), length=100):
    self.size = tuple(size)
    self.length = length
    self.count = 0
    self.done = False
    # name -> (space, constant fill); `count`, `reward` and the flags vary.
    self._constant = {
        'image': (Space(np.uint8, self.size + (3,)), 255),
        'vector': (Space(np.float32, (7,)), 0),
        'token': (Space(np.int32, (), 0, 256), 0),
        'float2d': (Space(np.float32, (4, 5)), 1),
        'int2d': (Space(np.int32, (2, 3), 0, 4), 1),
    }

  @property
  def obs_space(self):
    spaces = {name: space for name, (space, _) in self._constant.items()}
    spaces['count'] = Space(np.float32, (), 0, self.length)
    spaces['reward'] = Space(np.float32)
    for flag in ('is_first', 'is_last', 'is_terminal'):
      spaces[flag] = Space(bool)
    return spaces

  @property
  def act_space(self):
    return dict(
        reset=Space(bool),
        act_disc=Space(np.int32, (), 0, 5),
        act_cont=Space(np.float32, (6,)))

  def step(self, action):
    restart = bool(action['reset']) or self.done
    if restart:
      self.count, self.done = 0, False
    else:
      self.count += 1
      self.done = self.count >= self.length
    obs = {
        name: np.full(space.shape, fill, space.dtype)
        for name, (space, fill) in self._constant.items()}
    obs['count'] = np.float32(self.count)
    obs['reward'] = np.float32(0 if restart else 1)
    obs['is_first'] = restart
    obs['is_last'] = obs['is_terminal'] = (not restart) and self.done
    return obs
