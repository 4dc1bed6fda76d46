"""Numpy cart-pole (the classic control task behind `gym:CartPole-v1`,
BASELINE config 1) implementing the `Env` protocol; the reference reaches it
through `embodied/envs/from_gym.py`, gym is not available here.  Standard
dynamics: Euler integration at 50 Hz, force +-10 N, episode ends at |x| > 2.4,
|theta| > 12 degrees or after 500 steps.  The 4-vector observation is stored
under `vector`."""
import numpy as np

from ..core import base
from ..space import Space

GRAVITY, CART, POLE, HALF_LEN, FORCE, DT = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02


class CartPole(base.Env):

  def __init__(self, task='v1', seed=0, limit=500):
    self.rng = np.random.default_rng(seed)
    self.limit = limit
    self.state = np.zeros(4)
    self.t = 0
    self.done = True

  @property
  def obs_space(self):
    return {
        'vector': Space(np.float32, (4,)),
        'reward': Space(np.float32),
        'is_first': Space(bool),
        'is_last': Space(bool),
        'is_terminal': Space(bool),
    }

  @property
  def act_space(self):
    return {'reset': Space(bool), 'action': Space(np.int32, (), 0, 2)}

  def step(self, action):
    if bool(action['reset']) or self.done:
      self.state = self.rng.uniform(-0.05, 0.05, 4)
      self.t, self.done = 0, False
      return self._obs(0.0, is_first=True)
    x, v, th, w = self.state
    force = FORCE if int(action['action']) == 1 else -FORCE
    total = CART + POLE
    sin, cos = np.sin(th), np.cos(th)
    temp = (force + POLE * HALF_LEN * w * w * sin) / total
    alpha = (GRAVITY * sin - cos * temp) / (
        HALF_LEN * (4.0 / 3.0 - POLE * cos * cos / total))
    acc = temp - POLE * HALF_LEN * alpha * cos / total
    self.state = np.array([x + DT * v, v + DT * acc, th + DT * w, w + DT * alpha])
    self.t += 1
    fell = abs(self.state[0]) > 2.4 or abs(self.state[2]) > 12 * np.pi / 180
    self.done = bool(fell or self.t >= self.limit)
    return self._obs(1.0, is_last=self.done, is_terminal=bool(fell))

  def _obs(self, reward, is_first=False, is_last=False, is_terminal=False):
    return {
        'vector': self.state.astype(np.float32),
        'reward': np.float32(reward),
        'is_first': is_first,
        'is_last': is_last,
        'is_terminal': is_terminal,
    }
