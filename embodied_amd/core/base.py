"""The drop-in boundary: the duck-typed protocols every agent, env and stream
is written against (reference: embodied/core/base.py:1-73).  Method names,
argument order and the NotImplementedError texts (which spell the expected
signature) are what callers rely on, so they are kept; the classes are
generated from the signature tables below.
"""

AGENT_METHODS = {
    'init_train': ('batch_size', 'carry'),
    'init_report': ('batch_size', 'carry'),
    'init_policy': ('batch_size', 'carry'),
    'train': ('carry, data', 'carry, out, metrics'),
    'report': ('carry, data', 'carry, metrics'),
    'policy': ('carry, obs, mode', 'carry, act, out'),
    'stream': ('st', 'st'),
    'save': ('', 'data'),
    'load': ('data', 'None'),
}


def _unimplemented(name, params, returns):
  message = f'{name}({params}) -> {returns}'

  def method(self, *args, **kwargs):
    raise NotImplementedError(message)

  method.__name__ = name
  method.__doc__ = f'Expected signature: {message}'
  return method


class Agent:
  """policy / train / report / stream / init_* / save / load.  `policy`'s `out`
  keys must be disjoint from its `act` keys (driver.py:70-71); an
  `out['replay']` from `train` is fed to `Replay.update` (run/train.py:77-78)."""

  def __init__(self, obs_space, act_space, config):
    pass


for _name, (_params, _returns) in AGENT_METHODS.items():
  setattr(Agent, _name, _unimplemented(_name, _params, _returns))


class Env:
  """One environment instance.  `obs_space` must contain is_first, is_last and
  is_terminal (usually also reward and image); keys starting with 'log/' are
  shown to callbacks but neither to the agent nor to the replay; `act_space`
  must contain `reset`."""

  @property
  def obs_space(self):
    raise NotImplementedError('Returns: dict of spaces')

  @property
  def act_space(self):
    raise NotImplementedError('Returns: dict of spaces')

  def step(self, action):
    raise NotImplementedError('Returns: dict')

  def close(self):
    return None

  def __repr__(self):
    name = type(self).__name__
    return f'{name}(obs_space={self.obs_space}, act_space={self.act_space})'


class Stream:
  """An iterator whose position can be checkpointed."""

  def __iter__(self):
    return self

  def __next__(self):
    raise NotImplementedError

  def save(self):
    raise NotImplementedError

  def load(self, state):
    raise NotImplementedError
