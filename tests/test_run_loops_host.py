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
