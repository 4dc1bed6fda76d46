"""run.train end to end with a self-checking agent, the integration the
reference pins in embodied/tests/test_train.py:12-33 + tests/utils.py:8-104
(count continuity Driver -> Replay -> batch; train/report/save/load counts)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class CheckingAgent:
  """Asserts `obs['count']` continuity in policy (per env) and along T in every
  training batch; counts calls."""

  def __init__(self, obs_space, act_space):
    self.obs_space, self.act_space = obs_space, act_space
    self.policies = self.trains = self.reports = self.saves = self.loads = 0
    self.prev = None

  def init_policy(self, n):
    return ()

  init_train = init_report = init_policy

  def stream(self, st):
    return st

  def policy(self, carry, obs, mode='train'):
    count = np.asarray(obs['count'].cpu() if torch.is_tensor(obs['count']) else obs['count'])
    first = np.asarray(obs['is_first'].cpu() if torch.is_tensor(obs['is_first']) else obs['is_first'])
    if self.prev is not None:
      ok = np.where(first, count == 0, count == self.prev + 1)
      assert ok.all(), (self.prev, count, first)
    self.prev = count
    self.policies += 1
    n = len(first)
    act = {k: np.stack([v.sample() for _ in range(n)])
           for k, v in self.act_space.items() if k != 'reset'}
    return carry, act, {}

  def train(self, carry, data):
    count = data['count'].cpu().numpy() if torch.is_tensor(data['count']) else data['count']
    first = data['is_first'].cpu().numpy() if torch.is_tensor(data['is_first']) else data['is_first']
    for b in range(count.shape[0]):
      for t in range(1, count.shape[1]):
        assert first[b, t] or count[b, t] == count[b, t - 1] + 1
    assert 'consec' in data and 'stepid' in data
    self.trains += 1
    return carry, {}, {'loss': np.float32(self.trains)}

  def report(self, carry, data):
    self.reports += 1
    return carry, {'metric': np.float32(1)}

  def save(self):
    self.saves += 1
    return {'trains': self.trains}

  def load(self, data):
    self.loads += 1


def make_args(tmp_path, **kw):
  args = dict(
      logdir=str(tmp_path), batch_size=4, batch_length=8, train_ratio=32.0,
      log_every=0.05, report_every=0.05, save_every=0.05, envs=3, debug=True,
      from_checkpoint='', steps=600, consec_report=1, report_batches=1, device='cuda')
  args.update(kw)
  return types.SimpleNamespace(**args)


def run(tmp_path, agent_box, **kw):
  import embodied_amd as emb
  from embodied_amd.envs import dummy
  args = make_args(tmp_path, **kw)
  env0 = dummy.Dummy('disc', size=(8, 8), length=17)

  def make_agent():
    agent_box.append(CheckingAgent(env0.obs_space, env0.act_space))
    return agent_box[-1]

  make_replay = lambda: emb.Replay(
      length=args.batch_length + 1, capacity=500, chunksize=64,
      directory=tmp_path / 'replay', save_wait=True)
  make_env = lambda i: dummy.Dummy('disc', size=(8, 8), length=17 + i)
  make_stream = lambda replay, mode: emb.streams.Consec(
      emb.streams.Stateless(replay.sample, args.batch_size, mode),
      length=args.batch_length, consec=1, prefix=1, strict=True, contiguous=True)
  logger = emb.utils.Logger()
  emb.run.train(make_agent, make_replay, make_env, make_stream, lambda: logger, args)
  return logger, agent_box[-1]


def test_train_loop_counts_and_resume(tmp_path):
  box = []
  logger, agent = run(tmp_path, box)
  assert int(logger.step) >= 600
  # one policy call per vectorised step
  assert abs(agent.policies - int(logger.step) / 3) <= 4
  expected_trains = (int(logger.step) - 4 * 8) * 32.0 / (4 * 8)
  assert 0.7 * expected_trains <= agent.trains <= 1.1 * expected_trains + 2
  assert agent.reports >= 1
  assert agent.saves >= 2
  assert agent.loads == 0
  stats_seen = [r for r in logger.history if 'replay/items' in r]
  assert stats_seen and stats_seen[-1]['replay/items'] > 0
  # resume: checkpoint exists -> agent.load called once, replay restored
  logger2, agent2 = run(tmp_path, box, steps=700)
  assert agent2.loads == 1
  assert int(logger2.step) >= 700


class _PathLikeElements:
  """Stands in for `elements.Path` (ppo/main.py:190): not an os.PathLike, its
  str() is the path, `/` joins."""

  def __init__(self, path):
    self._path = str(path)

  def __truediv__(self, part):
    return _PathLikeElements(f'{self._path}/{part}')

  def __str__(self):
    return self._path


def test_train_with_factories_spelled_like_the_shipped_mains(tmp_path):
  """`make_replay` / `make_stream` / `wrap_env` as ppo/main.py:183-272 writes
  them, with `embodied` = this package: `embodied.replay.Replay(**kwargs)` over
  an elements-style directory, the three-way `embodied.replay.selectors.Mixture`
  with a recency table, `Stateless(bind(replay.sample, batch, mode))` under
  `Consec(..., contiguous=True)`, the wrapper chain of `wrap_env`."""
  from functools import partial as bind
  import embodied_amd as embodied
  from embodied_amd.envs import dummy
  config = types.SimpleNamespace(
      batch_size=4, batch_length=8, report_length=8, consec_train=1, consec_report=1, replay_context=1,
      logdir=str(tmp_path), replicas=1, replica=0,
      replay=types.SimpleNamespace(
          size=400, online=True, chunksize=64, recexp=1.0,
          fracs={'uniform': 0.5, 'priority': 0.3, 'recency': 0.2},
          prio=dict(exponent=0.8, maxfrac=0.5, initial=np.inf, zero_on_sample=True)))

  def make_replay(config, folder, mode='train'):
    batlen = config.batch_length if mode == 'train' else config.report_length
    consec = config.consec_train if mode == 'train' else config.consec_report
    capacity = config.replay.size if mode == 'train' else config.replay.size / 10
    length = consec * batlen + config.replay_context
    assert config.batch_size * length <= capacity
    directory = _PathLikeElements(config.logdir) / folder
    kwargs = dict(
        length=length, capacity=int(capacity), online=config.replay.online,
        chunksize=config.replay.chunksize, directory=directory)
    if config.replay.fracs['uniform'] < 1 and mode == 'train':
      recency = 1.0 / np.arange(1, capacity + 1) ** config.replay.recexp
      selectors = embodied.replay.selectors
      kwargs['selector'] = selectors.Mixture(dict(
          uniform=selectors.Uniform(),
          priority=selectors.Prioritized(**config.replay.prio),
          recency=selectors.Recency(recency),
      ), dict(config.replay.fracs))
    return embodied.replay.Replay(**kwargs)

  def wrap_env(env, config):
    for name, space in env.act_space.items():
      if not space.discrete:
        env = embodied.wrappers.NormalizeAction(env, name)
    env = embodied.wrappers.UnifyDtypes(env)
    env = embodied.wrappers.CheckSpaces(env)
    for name, space in env.act_space.items():
      if not space.discrete:
        env = embodied.wrappers.ClipAction(env, name)
    return env

  def make_stream(config, replay, mode):
    fn = bind(replay.sample, config.batch_size, mode)
    stream = embodied.streams.Stateless(fn)
    return embodied.streams.Consec(
        stream, length=config.batch_length if mode == 'train' else config.report_length,
        consec=config.consec_train if mode == 'train' else config.consec_report,
        prefix=config.replay_context, strict=(mode == 'train'), contiguous=True)

  box, replays = [], []
  env0 = wrap_env(dummy.Dummy('disc', size=(8, 8), length=17), config)

  def make_agent():
    box.append(CheckingAgent(env0.obs_space, env0.act_space))
    return box[-1]

  def replay_factory():
    replays.append(make_replay(config, 'replay'))
    return replays[-1]

  args = make_args(tmp_path, steps=500)
  logger = embodied.utils.Logger()
  embodied.run.train(
      make_agent, replay_factory, lambda i: wrap_env(dummy.Dummy('disc', size=(8, 8), length=17 + i), config),
      bind(make_stream, config), lambda: logger, args)
  agent = box[-1]
  assert int(logger.step) >= 500 and agent.trains > 50 and agent.reports >= 1
  assert list((tmp_path / 'replay').glob('*.npz'))            # the elements-style directory was used
  stream = make_stream(config, replays[-1], 'train')
  assert type(stream.source.fn).__name__ == 'method' and stream.source.args == (4, 'train')


def test_pretrain_from_the_replay_directory_of_a_run(tmp_path):
  """embodied/run/pretrain.py:8-96 on the path: `run.train` leaves its replay
  directory behind (chunk files); `run.pretrain` builds its three streams over a
  Replay loaded from that directory (`make_stream(None, mode)`) and takes
  train / report / eval batches from it -- the checking agent asserts count
  continuity along every window -- then resumes from its own checkpoint."""
  import embodied_amd as emb
  box = []
  run(tmp_path / 'collect', box)                  # 600 env steps, checkpoints on a 50 ms clock
  directory = tmp_path / 'collect' / 'replay'
  assert list(directory.glob('*.npz'))
  args = types.SimpleNamespace(
      logdir=str(tmp_path / 'pretrain'), steps=40, batch_size=4, batch_length=8, log_every=-1,
      report_every=-1, save_every=-1, consec_report=1, report_batches=2, replica=0,
      from_checkpoint='', from_checkpoint_regex=None)
  replays = []

  def make_stream(replay, mode):
    assert replay is None
    loaded = emb.Replay(length=args.batch_length + 1, capacity=500, chunksize=64, directory=directory,
                        seed={'train': 1, 'report': 2, 'eval': 3}[mode])
    loaded.load()
    assert len(loaded) > 0
    replays.append(loaded)
    return emb.streams.Consec(
        emb.streams.Stateless(loaded.sample, args.batch_size, mode),
        length=args.batch_length, consec=1, prefix=1, strict=True, contiguous=True)

  def make_model():
    from embodied_amd.envs import dummy
    env0 = dummy.Dummy('disc', size=(8, 8), length=17)
    box.append(CheckingAgent(env0.obs_space, env0.act_space))
    return box[-1]

  logger = emb.utils.Logger()
  emb.run.pretrain(make_model, make_stream, lambda: logger, args)
  model = box[-1]
  assert model.trains == 40 and int(logger.step) == 40 and model.loads == 0
  assert model.reports == 40 * 4                              # report + eval, two batches each, every step
  assert len(replays) == 3 and model.saves >= 40
  assert any('train/loss' in r for r in logger.history)
  logger2 = emb.utils.Logger()
  args.steps = 50
  emb.run.pretrain(make_model, make_stream, lambda: logger2, args)
  assert box[-1].loads == 1 and int(logger2.step) == 50


def test_actor_learner_split_respects_samples_per_insert(tmp_path):
  """Actor and learner threads on separate HIP streams, coupled by the
  SamplesPerInsert limiter: the realised sample/insert ratio stays near
  train_ratio / batch_length and batches stay consistent (the agent asserts
  count continuity on every batch, so a cross-stream race would trip it)."""
  import embodied_amd as emb
  from embodied_amd.envs import dummy
  args = make_args(tmp_path, steps=3000, train_ratio=16.0, log_every=0.2)
  env0 = dummy.Dummy('disc', size=(8, 8), length=17)
  box = []

  def make_agent():
    box.append(CheckingAgent(env0.obs_space, env0.act_space))
    return box[-1]

  replay_box = []
  def make_replay():
    replay_box.append(emb.Replay(length=args.batch_length + 1, capacity=400, chunksize=32))
    return replay_box[-1]

  make_stream = lambda replay, mode: emb.streams.Consec(
      emb.streams.Stateless(replay.sample, args.batch_size, mode),
      length=args.batch_length, consec=1, prefix=1, strict=True, contiguous=True)
  logger = emb.utils.Logger()
  counters = emb.run.actor_learner(
      make_agent, make_replay, lambda i: dummy.Dummy('disc', size=(8, 8), length=17 + i),
      make_stream, lambda: logger, args)
  agent = box[-1]
  assert int(logger.step) >= 3000
  assert agent.trains == counters['trains'] > 10
  # samples taken / steps inserted ~= train_ratio / batch_length (after warm-up)
  want = args.train_ratio / args.batch_length
  got = counters['trains'] * args.batch_size / int(logger.step)
  assert 0.6 * want <= got <= 1.2 * want, (got, want)
  assert replay_box[-1]._multistream


def test_train_eval_runs_both_drivers_and_replays(tmp_path):
  """run.train_eval (embodied/run/train_eval.py:10-18 signature): the train
  Driver feeds the train Replay and the learner, the eval Driver runs whole
  episodes in mode='eval' into its own Replay, both report streams are drawn,
  checkpoints hold both replays."""
  import embodied_amd as emb
  from embodied_amd.envs import dummy
  args = make_args(tmp_path, steps=500, eval_every=0.05, eval_envs=2, eval_eps=2)
  env0 = dummy.Dummy('disc', size=(8, 8), length=17)
  modes = []

  class Agent(CheckingAgent):
    def policy(self, carry, obs, mode='train'):
      modes.append((mode, len(obs['is_first'])))
      self.prev = None              # two drivers interleave: continuity is checked in train()
      return super().policy(carry, obs, mode)

  agent = Agent(env0.obs_space, env0.act_space)
  replays = {}
  def make_replay(name, capacity):
    replays[name] = emb.Replay(
        length=args.batch_length + 1, capacity=capacity, chunksize=64,
        directory=tmp_path / name, save_wait=True)
    return replays[name]
  make_stream = lambda replay, mode: emb.streams.Consec(
      emb.streams.Stateless(replay.sample, args.batch_size, mode),
      length=args.batch_length, consec=1, prefix=1, strict=True, contiguous=True)
  logger = emb.utils.Logger()
  emb.run.train_eval(
      lambda: agent, lambda: make_replay('replay', 500), lambda: make_replay('eval_replay', 100),
      lambda i: dummy.Dummy('disc', size=(8, 8), length=17 + i),
      lambda i: dummy.Dummy('disc', size=(8, 8), length=11), make_stream, lambda: logger, args)
  assert int(logger.step) >= 500
  train_calls = [n for mode, n in modes if mode == 'train']
  eval_calls = [n for mode, n in modes if mode == 'eval']
  assert set(train_calls) == {3} and set(eval_calls) == {2}
  assert abs(len(train_calls) - int(logger.step) / 3) <= 4          # only train steps count
  assert len(eval_calls) >= 2 * 12                                  # >= 2 whole episodes per evaluation
  assert agent.trains > 10 and agent.reports >= 2                   # eval + train report streams
  assert len(replays['replay']) > 0 and len(replays['eval_replay']) > 0
  assert agent.saves >= 1
  assert list((tmp_path / 'replay').glob('*.npz')) and list((tmp_path / 'eval_replay').glob('*.npz'))
  keys = set().union(*logger.history)
  assert any(k.startswith('eval/') for k in keys) and any(k.startswith('report/') for k in keys)
