"""Run-loop pieces that need no GPU: the vectorised episode statistics against a
per-env restatement of the reference's `logfn` callback
(embodied/run/train.py:31-54), and `run.eval_only` end to end on host envs
(embodied/run/eval_only.py:9-74; the reference's own pattern for such tests is
embodied/tests/test_train.py:12-33: Dummy envs + a counting agent)."""
import types

import numpy as np

import embodied_amd as emb
from embodied_amd.run.stats import EpisodeStats


class Sink:
  def __init__(self):
    self.rows = []

  def add(self, mapping, prefix=None):
    self.rows.append({(f'{prefix}/{k}' if prefix else k): float(v) for k, v in mapping.items()})


class PerEnvEpisodes:
  """What the reference's per-env `logfn` callback reports (run/train.py:31-54),
  written as one plain accumulator per env: an episode starts at `is_first`,
  every step adds its reward to the score and one to the length and remembers the
  reward, every scalar `log/*` value is tracked for avg / max / sum; at `is_last`
  score and length go to the logger under `episode/`, and the episode's
  reward_rate (share of consecutive reward pairs that differ by >= 0.01, only
  with more than one step) plus the log aggregates go to the episode statistics.
  (The uint8 image stacks kept for worker 0's video are not part of the path.)"""

  def __init__(self, logger, epstats, n):
    self.logger, self.epstats = logger, epstats
    self.envs = [self._fresh() for _ in range(n)]

  @staticmethod
  def _fresh():
    return {'rewards': [], 'logs': {}}

  def __call__(self, tran, worker):
    if tran['is_first']:
      self.envs[worker] = self._fresh()
    env = self.envs[worker]
    env['rewards'].append(float(tran['reward']))
    for key, value in tran.items():
      if key.startswith('log/'):
        assert np.ndim(value) == 0
        env['logs'].setdefault(key, []).append(float(value))
    if not tran['is_last']:
      return
    rewards = np.asarray(env['rewards'])
    self.logger.add({'score': rewards.sum(), 'length': len(rewards)}, prefix='episode')
    result = {}
    for key, values in env['logs'].items():
      result[f'{key}/avg'] = np.mean(values)
      result[f'{key}/max'] = np.max(values)
      result[f'{key}/sum'] = np.sum(values)
    if len(rewards) > 1:
      result['reward_rate'] = (np.abs(np.diff(rewards)) >= 0.01).mean()
    self.epstats.add(result)


def test_episode_stats_equal_per_env_logfn():
  n, steps = 5, 400
  gen = np.random.default_rng(0)
  ours_log, ours_ep = Sink(), Sink()
  want_log, want_ep = Sink(), Sink()
  stats = EpisodeStats(ours_log, ours_ep)
  logfn = PerEnvEpisodes(want_log, want_ep, n)
  is_last = np.zeros(n, bool)
  for t in range(steps):
    is_first = is_last.copy() if t else np.ones(n, bool)
    is_last = gen.random(n) < 0.07
    reward = np.where(gen.random(n) < 0.5, 0.0, gen.standard_normal(n)).astype(np.float32)
    trans = {
        'reward': reward, 'is_first': is_first, 'is_last': is_last,
        'image': gen.integers(0, 255, (n, 4, 4, 3), dtype=np.uint8),
        'log/height': gen.standard_normal(n).astype(np.float32),
        'log/coins': gen.integers(0, 3, n).astype(np.float32)}
    stats.on_batch(trans, np.arange(n))
    for i in range(n):
      logfn({k: v[i] for k, v in trans.items()}, i)
  assert len(ours_log.rows) == len(want_log.rows) > 10
  for got, want in zip(ours_log.rows + ours_ep.rows, want_log.rows + want_ep.rows):
    assert set(got) == set(want), (got, want)
    for k in want:
      np.testing.assert_allclose(got[k], want[k], rtol=1e-6, atol=1e-6, err_msg=k)
  assert any('reward_rate' in row for row in ours_ep.rows)
  assert {'log/height/avg', 'log/height/max', 'log/height/sum'} <= set(ours_ep.rows[0])


class CountingAgent:
  def __init__(self, act_space):
    self.act_space = act_space
    self.modes, self.loads = [], 0

  def init_policy(self, n):
    return np.zeros(n, np.int64)

  def policy(self, carry, obs, mode='train'):
    self.modes.append(mode)
    n = len(obs['is_first'])
    act = {k: np.stack([v.sample() for _ in range(n)])
           for k, v in self.act_space.items() if k != 'reset'}
    return carry + 1, act, {}

  def load(self, data):
    self.loads += 1


def test_eval_only_steps_envs_in_eval_mode_and_logs_episodes(tmp_path):
  from embodied_amd.envs import dummy
  env0 = dummy.Dummy('disc', size=(8, 8), length=9)
  agent = CountingAgent(env0.act_space)
  logger = emb.utils.Logger()
  args = types.SimpleNamespace(
      logdir=str(tmp_path), envs=3, debug=True, steps=120, log_every=-1, from_checkpoint='')
  emb.run.eval_only(
      lambda: agent, lambda i: dummy.Dummy('disc', size=(8, 8), length=9 + i), lambda: logger, args)
  assert int(logger.step) >= 120 and int(logger.step) % 3 == 0
  assert set(agent.modes) == {'eval'} and len(agent.modes) == int(logger.step) // 3
  keys = set().union(*logger.history)
  assert {'episode/score', 'episode/length', 'fps/policy'} <= keys
  lengths = [r['episode/length'] for r in logger.history if 'episode/length' in r]
  assert lengths and all(9 <= int(x) <= 12 for x in lengths)     # Dummy: `length` steps + the reset step


class _CountingStream:
  """A restorable stream of batches {'count': [k, k + 1, ...]}."""

  def __init__(self, mode):
    self.mode, self.k = mode, 0

  def __iter__(self):
    return self

  def __next__(self):
    self.k += 1
    return {'count': np.arange(self.k, self.k + 4), 'mode': self.mode}

  def save(self):
    return self.k

  def load(self, k):
    self.k = k


class _CountingModel:
  """The learner side of the reference's self-checking TestAgent
  (embodied/tests/utils.py:8-104): counts calls, checks what it is fed."""

  def __init__(self):
    self.trains, self.reports, self.loaded, self.seen = 0, {'report': 0, 'eval': 0}, None, []

  def stream(self, st):
    return st

  def init_train(self, batch_size):
    return ('train', batch_size)

  def init_report(self, batch_size):
    return ('report', batch_size)

  def train(self, carry, batch):
    assert carry == ('train', 4) and batch['mode'] == 'train'
    self.seen.append(int(batch['count'][0]))
    self.trains += 1
    return carry, {}, {'loss': 1.0 / self.trains}

  def report(self, carry, batch):
    assert carry == ('report', 4) and batch['mode'] in ('report', 'eval')
    self.reports[batch['mode']] += 1
    return carry, {'score': float(batch['count'][0])}

  def save(self):
    return {'trains': self.trains}

  def load(self, data, regex=None):
    self.loaded = (data, regex)
    self.trains = data['trains']


def _pretrain_args(tmp_path, **over):
  base = dict(
      logdir=str(tmp_path), steps=25, batch_size=4, batch_length=8, log_every=-1, report_every=-1,
      save_every=-1, consec_report=2, report_batches=3, replica=0, from_checkpoint='',
      from_checkpoint_regex=None)
  base.update(over)
  return types.SimpleNamespace(**base)


def test_pretrain_trains_reports_evaluates_and_checkpoints(tmp_path):
  """embodied/run/pretrain.py:8-96: `steps` train steps from the train stream,
  consec_report * report_batches report and eval batches per report, train
  metrics / fps on the log schedule, a checkpoint holding step, model and the
  three streams; a second call resumes from it."""
  model = _CountingModel()
  made = {}

  def make_stream(replay, mode):
    assert replay is None
    made[mode] = _CountingStream(mode)
    return made[mode]

  logger = emb.utils.Logger()
  emb.run.pretrain(lambda: model, make_stream, lambda: logger, _pretrain_args(tmp_path))
  assert model.trains == 25 and model.seen == list(range(1, 26)) and int(logger.step) == 25
  assert model.reports == {'report': 25 * 6, 'eval': 25 * 6}      # every step is due with -1
  rows = logger.history
  assert any('train/loss' in r for r in rows) and any('report/score' in r for r in rows)
  assert any('eval/score' in r for r in rows) and any('fps' in r for r in rows)
  assert (tmp_path / 'checkpoint.pkl').exists()
  # resume: the counter, the model and the stream positions come back
  again, streams2 = _CountingModel(), {}

  def make_stream2(replay, mode):
    streams2[mode] = _CountingStream(mode)
    return streams2[mode]

  logger2 = emb.utils.Logger()
  emb.run.pretrain(lambda: again, make_stream2, lambda: logger2, _pretrain_args(tmp_path, steps=30))
  assert again.loaded[0] == {'trains': 25} and again.trains == 30 and int(logger2.step) == 30
  assert again.seen == list(range(26, 31))


def test_pretrain_starts_from_another_checkpoint(tmp_path):
  import pickle
  source = tmp_path / 'other.pkl'
  source.write_bytes(pickle.dumps({'model': {'trains': 100}}))
  model = _CountingModel()
  emb.run.pretrain(
      lambda: model, lambda replay, mode: _CountingStream(mode), lambda: emb.utils.Logger(),
      _pretrain_args(tmp_path / 'run', steps=3, from_checkpoint=str(source), from_checkpoint_regex='enc/.*',
                     report_every=0))
  assert model.loaded == ({'trains': 100}, 'enc/.*') and model.trains == 103
  assert model.reports == {'report': 0, 'eval': 0}


def test_global_clock_without_a_process_group_is_a_local_clock():
  always, never = emb.GlobalClock(-1), emb.GlobalClock(0)
  assert always() and always(skip=False) and not always(skip=True) and not never()
  timed = emb.GlobalClock(1000.0, first=True)
  assert timed() and not timed()


def _reference_logfn(logger, epstats):
  """The reference's own per-env `logfn` (the function nested in
  embodied/run/train.py:31-54), taken out of the file's syntax tree at run time
  and compiled with the names it closes over: `episodes` / `epstats` are this
  package's `utils.Agg` (the stand-in for the un-vendored elements.Agg, SURVEY
  Appendix A), `logger` the test's sink."""
  import ast
  import collections
  from oracle import refload
  path = refload.REFERENCE / 'embodied' / 'run' / 'train.py'
  tree = ast.parse(path.read_text(), filename=str(path))
  train = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'train')
  node = next(n for n in ast.walk(train) if isinstance(n, ast.FunctionDef) and n.name == 'logfn')
  node.decorator_list = []                      # (@elements.timer.section: timing only)
  namespace = {
      'np': np, 'episodes': collections.defaultdict(emb.utils.Agg), 'logger': logger, 'epstats': epstats}
  exec(compile(ast.Module(body=[node], type_ignores=[]), str(path), 'exec'), namespace)
  return namespace['logfn']


def test_episode_stats_equal_the_references_own_logfn():
  """Build container only: the vectorised `EpisodeStats` against the reference's
  `logfn` itself, executed from its source (not a restatement), on random
  episodes with images, rewards and scalar `log/*` values."""
  import pytest
  from oracle import refload
  if not refload.available():
    pytest.skip('no /root/reference here')
  n, steps = 5, 400
  gen = np.random.default_rng(1)

  class ScalarSink(Sink):
    # the reference keeps worker 0's image stacks in the episode result (its
    # video logging, `policy_*`): not a statistic, not kept
    def add(self, mapping, prefix=None):
      super().add({k: v for k, v in mapping.items() if np.ndim(v) == 0}, prefix)

  ours_log, ours_ep = Sink(), Sink()
  want_log, want_ep = ScalarSink(), ScalarSink()
  stats = EpisodeStats(ours_log, ours_ep)
  logfn = _reference_logfn(want_log, want_ep)
  is_last = np.zeros(n, bool)
  for t in range(steps):
    is_first = is_last.copy() if t else np.ones(n, bool)
    is_last = gen.random(n) < 0.07
    reward = np.where(gen.random(n) < 0.5, 0.0, gen.standard_normal(n)).astype(np.float32)
    trans = {
        'reward': reward, 'is_first': is_first, 'is_last': is_last,
        'image': gen.integers(0, 255, (n, 4, 4, 3), dtype=np.uint8),
        'log/height': gen.standard_normal(n).astype(np.float32),
        'log/coins': gen.integers(0, 3, n).astype(np.float32)}
    stats.on_batch(trans, np.arange(n))
    for i in range(n):
      logfn({k: v[i] for k, v in trans.items()}, i)
  stats.flush()
  assert len(want_log.rows) > 20 and len(want_ep.rows) == len(want_log.rows)
  key = lambda row: sorted(row.items())
  for ours, want in ((ours_log, want_log), (ours_ep, want_ep)):
    a, b = sorted(map(key, ours.rows)), sorted(map(key, want.rows))
    assert len(a) == len(b)
    for x, y in zip(a, b):
      assert [k for k, _ in x] == [k for k, _ in y]
      np.testing.assert_allclose([v for _, v in x], [v for _, v in y], rtol=1e-6, atol=1e-6)
