"""Per-env wrappers the shipped agents put around every env before the Driver
sees it (reference: embodied/core/wrappers.py; call site `wrap_env`,
ppo/main.py:249-258 and dreamerv3/main.py).

Host-side glue on single steps, not part of the HBM path; provided so that
`make_env` functions written for the reference run unchanged.  Each wrapper is
a pair of hooks around the inner env's `step` — `_down(action)` rewrites the
action dict on its way in, `_up(obs)` the observation on its way out — plus
optional overrides of the two space dicts.
"""
import numpy as np

from ..space import Space


class Wrapper:
  """Delegates everything it does not define to the wrapped env."""

  def __init__(self, env):
    self.env = env

  def __getattr__(self, name):
    # Only reached for names not found on the wrapper itself.
    if name.startswith('__'):
      raise AttributeError(name)
    try:
      return getattr(self.env, name)
    except AttributeError:
      raise ValueError(name)     # wrappers.py:19-25 turns a miss into ValueError

  def __len__(self):
    return len(self.env)

  def __bool__(self):
    return bool(self.env)

  def _down(self, action):
    return action

  def _up(self, obs):
    return obs

  def step(self, action):
    return self._up(self.env.step(self._down(action)))


class TimeLimit(Wrapper):
  """Ends the episode after `duration` steps (wrappers.py:28-54).  With
  `reset=False` the inner env is not reset at the limit: the next step is
  flagged `is_first` instead."""

  def __init__(self, env, duration, reset=True):
    super().__init__(env)
    self._duration = duration
    self._hard = reset
    self._elapsed = 0
    self._over = False

  def step(self, action):
    if action['reset'] or self._over:
      self._elapsed, self._over = 0, False
      action.update(reset=bool(self._hard))      # in place, as the reference
      obs = self.env.step(action)
      if not self._hard:
        obs['is_first'] = True
      return obs
    self._elapsed += 1
    obs = self.env.step(action)
    if self._duration and self._elapsed >= self._duration:
      obs['is_last'] = True
    self._over = obs['is_last']
    return obs


class ActionRepeat(Wrapper):
  """Applies every action `repeat` times and sums the rewards
  (wrappers.py:57-73); stops early at an episode end; resets pass through."""

  def __init__(self, env, repeat):
    super().__init__(env)
    self._repeat = repeat

  def step(self, action):
    if action['reset']:
      return self.env.step(action)
    total, obs = 0.0, None
    for _ in range(self._repeat):
      obs = self.env.step(action)
      total += obs['reward']
      if obs['is_last'] or obs['is_terminal']:
        break
    obs['reward'] = np.float32(total)
    return obs


class ClipAction(Wrapper):
  """Clips one action entry into [low, high] (wrappers.py:76-86)."""

  def __init__(self, env, key='action', low=-1, high=1):
    super().__init__(env)
    self._key, self._low, self._high = key, low, high

  def _down(self, action):
    return {**action, self._key: np.clip(action[self._key], self._low, self._high)}


class NormalizeAction(Wrapper):
  """Presents a bounded continuous action as [-1, 1] and maps it back to the
  env's own range; dimensions without finite bounds pass through
  (wrappers.py:89-110)."""

  def __init__(self, env, key='action'):
    super().__init__(env)
    self._key = key
    inner = env.act_space[key]
    self._bounded = np.isfinite(inner.low) & np.isfinite(inner.high)
    self._low = np.where(self._bounded, inner.low, -1)
    self._high = np.where(self._bounded, inner.high, 1)
    ones = np.ones_like(self._low)
    self._outer = Space(
        np.float32, inner.shape,
        np.where(self._bounded, -ones, self._low),
        np.where(self._bounded, ones, self._high))

  @property
  def act_space(self):
    return {**self.env.act_space, self._key: self._outer}

  def _down(self, action):
    unit = action[self._key]
    scaled = (unit + 1) / 2 * (self._high - self._low) + self._low
    return {**action, self._key: np.where(self._bounded, scaled, unit)}


def _unified(dtype):
  """The three dtypes that cross the boundary (wrappers.py:228-241):
  floats -> float32, uint8 stays, other integers -> int32, rest unchanged."""
  if np.issubdtype(dtype, np.floating):
    return np.dtype(np.float32)
  if np.issubdtype(dtype, np.uint8):
    return np.dtype(np.uint8)
  if np.issubdtype(dtype, np.integer):
    return np.dtype(np.int32)
  return np.dtype(dtype)


class UnifyDtypes(Wrapper):
  """Outer spaces use float32 / uint8 / int32; actions are cast to the inner
  env's dtypes on the way in, observations to the unified ones on the way out
  (wrappers.py:204-241)."""

  def __init__(self, env):
    super().__init__(env)
    self._inner_act = {k: s.dtype for k, s in env.act_space.items()}
    self._outer_obs = {k: _unified(s.dtype) for k, s in env.obs_space.items()}
    self._obs_space = {
        k: Space(self._outer_obs[k], s.shape, s.low, s.high)
        for k, s in env.obs_space.items()}
    self._act_space = {
        k: Space(_unified(s.dtype), s.shape, s.low, s.high)
        for k, s in env.act_space.items()}

  @property
  def obs_space(self):
    return self._obs_space

  @property
  def act_space(self):
    return self._act_space

  def _down(self, action):
    action = dict(action)
    for key, dtype in self._inner_act.items():
      action[key] = np.asarray(action[key], dtype)
    return action

  def _up(self, obs):
    for key, dtype in self._outer_obs.items():
      obs[key] = np.asarray(obs[key], dtype)
    return obs


class CheckSpaces(Wrapper):
  """Raises when an action or observation entry leaves its declared space
  (wrappers.py:244-270): TypeError for foreign value types, ValueError for a
  wrong dtype kind, shape or range."""

  _PLAIN = (np.ndarray, np.generic, list, tuple, int, float, bool)

  def __init__(self, env):
    shared = env.obs_space.keys() & env.act_space.keys()
    assert not shared, (env.obs_space.keys(), env.act_space.keys())
    super().__init__(env)

  def _verify(self, values, spaces):
    for key, value in values.items():
      if not isinstance(value, self._PLAIN):
        raise TypeError(f'Invalid type {type(value)} for key {key}.')
      space = spaces[key]
      if value in space:
        continue
      array = np.array(value)
      raise ValueError(
          f"Value for '{key}' with dtype {array.dtype}, shape {array.shape}, "
          f"lowest {np.min(value)}, highest {np.max(value)} is not in {space}.")

  def _down(self, action):
    self._verify(action, self.env.act_space)
    return action

  def _up(self, obs):
    self._verify(obs, self.env.obs_space)
    return obs


class DiscretizeAction(Wrapper):
  """A (dims,) continuous action becomes (dims,) bin indices over
  linspace(-1, 1, bins) (wrappers.py:273-288)."""

  def __init__(self, env, key='action', bins=5):
    super().__init__(env)
    self._key = key
    (self._dims,) = env.act_space[key].shape
    self._values = np.linspace(-1, 1, bins)

  @property
  def act_space(self):
    space = Space(np.int32, self._dims, 0, len(self._values))
    return {**self.env.act_space, self._key: space}

  def _down(self, action):
    return {**action, self._key: np.take(self._values, action[self._key])}
