"""The drop-in boundary: the duck-typed protocols every agent, env and stream
is written against (reference: embodied/core/base.py:1-73).  Signatures and
error behaviour (NotImplementedError naming the expected signature) are kept so
code written for the reference runs unchanged."""


def _missing(signature):
  return NotImplementedError(signature)


class Agent:
  """policy/train/report/stream/init_*/save/load (base.py:1-31)."""

  def __init__(self, obs_space, act_space, config):
    pass

  def init_train(self, batch_size):
    raise _missing('init_train(batch_size) -> carry')

  def init_report(self, batch_size):
    raise _missing('init_report(batch_size) -> carry')

  def init_policy(self, batch_size):
    raise _missing('init_policy(batch_size) -> carry')

  def train(self, carry, data):
    raise _missing('train(carry, data) -> carry, out, metrics')

  def report(self, carry, data):
    raise _missing('report(carry, data) -> carry, metrics')

  def policy(self, carry, obs, mode):
    raise _missing('policy(carry, obs, mode) -> carry, act, out')

  def stream(self, st):
    raise _missing('stream(st) -> st')

  def save(self):
    raise _missing('save() -> data')

  def load(self, data):
    raise _missing('load(data) -> None')


class Env:
  """step/obs_space/act_space/close (base.py:34-58).  Observations carry
  is_first, is_last, is_terminal (and usually reward); keys starting with
  'log/' bypass agent and replay; the action space contains 'reset'."""

  def __repr__(self):
    return (f'{type(self).__name__}(obs_space={self.obs_space}, '
            f'act_space={self.act_space})')

  @property
  def obs_space(self):
    raise _missing('Returns: dict of spaces')

  @property
  def act_space(self):
    raise _missing('Returns: dict of spaces')

  def step(self, action):
    raise _missing('Returns: dict')

  def close(self):
    pass


class Stream:
  """Iterator with save()/load(state) (base.py:61-73)."""

  def __iter__(self):
    return self

  def __next__(self):
    raise NotImplementedError

  def save(self):
    raise NotImplementedError

  def load(self, state):
    raise NotImplementedError
