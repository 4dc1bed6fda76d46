"""Driver episode bookkeeping, re-expressing what the reference pins in
embodied/tests/test_driver.py:9-112 (host mode: numpy, no GPU).  Where the
reference test is stale — it reads tran['reset'], which the current Driver no
longer puts in the transition (driver.py:74-76) — the check is made on the
action the env receives instead."""
from functools import partial as bind

import numpy as np
import pytest

import embodied_amd as emb
from embodied_amd.envs import dummy


class Taped(dummy.Dummy):
  """The dummy env with a tape of what it was handed."""

  def __init__(self, length, tape):
    super().__init__('disc', length=length)
    self.tape = tape

  def step(self, action):
    self.tape.append({k: np.array(v) for k, v in action.items()})
    return super().step(action)


class Session:
  """One Driver run over `n` dummy envs of `length` steps with a policy that
  numbers its calls: keeps what every party saw -- the envs' inputs, the
  policy's inputs and carries, the transitions handed to the callbacks."""

  def __init__(self, n=1, length=10, parallel=False, **run):
    self.n, self.length = n, length
    self.handed = [[] for _ in range(n)]          # per env: the action dicts it received
    self.asked, self.carries, self.trans = [], [], [[] for _ in range(n)]
    makers = [bind(Taped, length, self.handed[i]) for i in range(n)] if not parallel else [
        bind(dummy.Dummy, 'disc', length=length) for _ in range(n)]
    driver = emb.Driver(makers, parallel=parallel)
    driver.reset(lambda count: 0)
    driver.on_step(lambda tran, worker: self.trans[worker].append({k: np.array(v) for k, v in tran.items()}))
    driver(self.policy, **run)
    driver.close()

  def policy(self, carry, obs, **kw):
    self.asked.append({k: np.array(v) for k, v in obs.items()})
    self.carries.append(carry)
    acts = {'act_disc': np.full(self.n, 1 + carry % 3, np.int32),
            'act_cont': np.full((self.n, 6), 0.25 * (1 + carry % 3), np.float32)}
    return carry + 1, acts, {}

  def column(self, worker, key):
    return np.array([t[key] for t in self.trans[worker]])


@pytest.mark.parametrize('parallel', [False, True])
@pytest.mark.parametrize('length,episodes', [(10, 1), (10, 2), (3, 4)])
def test_episodes_are_framed_by_their_flags(parallel, length, episodes):
  """An episode of a `length`-step env is length + 1 transitions (the reset
  observation comes first): is_first on the first only, is_last on the last
  only -- test_driver.py:9-44 as one statement over several shapes."""
  run = Session(1, length, parallel, episodes=episodes)
  first, last = run.column(0, 'is_first'), run.column(0, 'is_last')
  span = length + 1
  assert len(first) == episodes * span
  where = np.arange(len(first)) % span
  assert np.array_equal(first, where == 0)
  assert np.array_equal(last, where == span - 1)


@pytest.mark.parametrize('n,length', [(1, 5), (3, 4)])
def test_the_step_that_ends_an_episode_stores_no_action_and_asks_for_a_reset(n, length):
  """driver.py:72-76: where is_last is set the stored action is zero and the
  env's next input carries reset (the reference's test reads tran['reset'],
  which the current Driver no longer stores: checked on what the env receives)."""
  run = Session(n, length, steps=3 * (length + 1) * n)
  for w in range(n):
    last = run.column(w, 'is_last')
    disc, cont = run.column(w, 'act_disc'), run.column(w, 'act_cont')
    assert last.sum() == 3
    assert (disc[last] == 0).all() and (cont[last] == 0).all()
    assert (disc[~last] != 0).all() and (cont[~last] != 0).all()      # the policy never says zero
    assert 'reset' not in run.trans[w][0]
    resets = np.array([bool(a['reset']) for a in run.handed[w]])
    # the very first input resets; after that: exactly the inputs that follow an is_last
    assert resets[0] and np.array_equal(resets[1:len(last)], last[:len(resets) - 1][:len(last) - 1])
    # what the env is handed on step t + 1 is what was stored on step t
    for t in range(len(last) - 1):
      assert np.array_equal(run.handed[w][t + 1]['act_disc'], disc[t])


def test_the_policy_sees_the_stored_observations_and_its_own_carry():
  """test_driver.py:46-68: the policy's inputs are the transitions'
  observations, stacked over the envs, and its carry comes back to it."""
  run = Session(2, 10, episodes=4)              # (episodes are counted over both envs)
  assert run.carries == list(range(len(run.asked)))           # reset's carry, then the policy's own
  steps = len(run.asked)
  assert steps == len(run.trans[0]) == len(run.trans[1])
  for t, obs in enumerate(run.asked):
    for w in range(2):
      for key in ('is_first', 'is_last', 'is_terminal', 'reward'):
        assert np.array_equal(obs[key][w], run.trans[w][t][key]), (t, w, key)
  assert run.asked[0]['is_first'].all() and not run.asked[1]['is_first'].any()
  assert run.asked[10]['is_last'].all() and run.asked[11]['is_first'].all()


def make_env(length=10):
  return dummy.Dummy('disc', length=length)


def make_agent():
  env = make_env()
  agent = emb.RandomAgent(env.obs_space, env.act_space)
  env.close()
  return agent


def test_unexpected_reset_mid_episode():
  class UnexpectedReset:
    """is_first without a preceding is_last."""
    def __init__(self, env, when):
      self.env, self.when, self.count = env, when, 0
    act_space = property(lambda self: self.env.act_space)
    obs_space = property(lambda self: self.env.obs_space)
    def step(self, action):
      if self.count == self.when:
        action = {**action, 'reset': np.ones_like(action['reset'])}
      self.count += 1
      return self.env.step(action)
    def close(self):
      pass

  env = UnexpectedReset(make_env(length=4), when=3)
  agent = make_agent()
  driver = emb.Driver([lambda: env], parallel=False)
  driver.reset(agent.init_policy)
  steps = []
  driver.on_step(lambda tran, _: steps.append(tran))
  driver(agent.policy, episodes=1)
  assert len(steps) == 8
  cols = {k: np.array([x[k] for x in steps]) for k in steps[0]}
  assert (cols['is_first'] == [1, 0, 0, 1, 0, 0, 0, 0]).all()
  assert (cols['is_last'] == [0, 0, 0, 0, 0, 0, 0, 1]).all()


def test_out_keys_must_not_shadow_actions():
  agent = make_agent()
  driver = emb.Driver([make_env], parallel=False)
  driver.reset(agent.init_policy)
  def policy(carry, obs):
    _, act, _ = agent.policy(carry, obs)
    return carry, act, {'act_disc': np.zeros(1)}
  with pytest.raises(AssertionError):
    driver(policy, steps=1)


def test_log_keys_reach_callbacks_but_not_policy():
  class Logging(dummy.Dummy):
    def step(self, action):
      obs = super().step(action)
      obs['log/extra'] = np.float32(7)
      return obs
  seen, trans = [], []
  driver = emb.Driver([lambda: Logging('disc')], parallel=False)
  agent = make_agent()
  driver.reset(agent.init_policy)
  def policy(carry, obs):
    seen.append(set(obs))
    return agent.policy(carry, obs)
  driver.on_step(lambda tran, _: trans.append(set(tran)))
  driver(policy, steps=3)
  assert all('log/extra' not in s for s in seen)
  assert all('log/extra' in t for t in trans)


@pytest.mark.parametrize('shared_obs', [True, False])
def test_parallel_workers_match_serial(shared_obs):
  """Env processes (shared-memory obs slab or pickled pipes) produce exactly
  the transitions of the in-process loop."""
  from tests import scenarios

  def run(**kw):
    fns = [bind(scenarios.ScriptEnv, i, 3 + i) for i in range(3)]
    driver = emb.Driver(fns, **kw)
    log = []
    driver.on_step(lambda tran, w: log.append((w, {k: np.array(v) for k, v in tran.items()})))
    driver.reset(lambda n: 0)
    def policy(carry, obs):
      n = len(obs['is_first'])
      act = {'act_disc': (np.arange(n) + carry).astype(np.int32),
             'act_cont': np.full((n, 3), carry, np.float32)}
      return carry + 1, act, {}
    driver(policy, steps=36)
    driver.close()
    return log

  serial = run(parallel=False)
  par = run(parallel=True, shared_obs=shared_obs)
  assert len(serial) == len(par) == 36
  for (w0, t0), (w1, t1) in zip(serial, par):
    assert w0 == w1 and set(t0) == set(t1)
    for k in t0:
      assert np.array_equal(t0[k], t1[k]) and t0[k].dtype == t1[k].dtype, k


@pytest.mark.parametrize('per_worker', [2, 3, 5])
def test_several_envs_per_worker_process_match_serial(per_worker):
  """Driver(envs_per_worker=K): worker w of W steps envs w, w + W, w + 2W, ... one after the
  other (for hosts whose CPU budget is smaller than the env count); the
  transitions are the in-process loop's, also when N is no multiple of K."""
  from tests import scenarios

  def run(**kw):
    fns = [bind(scenarios.ScriptEnv, i, 3 + i) for i in range(7)]
    driver = emb.Driver(fns, **kw)
    log = []
    driver.on_step(lambda tran, w: log.append((w, {k: np.array(v) for k, v in tran.items()})))
    driver.reset(lambda n: 0)
    def policy(carry, obs):
      n = len(obs['is_first'])
      act = {'act_disc': (np.arange(n) + carry).astype(np.int32),
             'act_cont': np.full((n, 3), carry, np.float32)}
      return carry + 1, act, {}
    driver(policy, steps=70)
    procs = len(getattr(driver, 'procs', []))
    driver.close()
    return log, procs

  serial, _ = run(parallel=False)
  par, procs = run(parallel=True, envs_per_worker=per_worker)
  assert procs == -(-7 // per_worker)
  assert len(serial) == len(par) == 70
  for (w0, t0), (w1, t1) in zip(serial, par):
    assert w0 == w1 and set(t0) == set(t1)
    for k in t0:
      assert np.array_equal(t0[k], t1[k]) and t0[k].dtype == t1[k].dtype, k
  with pytest.raises(ValueError):
    emb.Driver([bind(scenarios.ScriptEnv, 0, 3)] * 2, parallel=True, envs_per_worker=2, shared_obs=False)


def test_errors_and_extras_with_several_envs_per_worker():
  agent = make_agent()
  driver = emb.Driver([lambda: _Flaky('disc', length=10)] * 4, parallel=True, envs_per_worker=2)
  assert driver._fast and len(driver.procs) == 2
  driver.reset(agent.init_policy)
  seen = []
  driver.on_step(lambda tran, w: seen.append(float(tran['log/steps'])))
  with pytest.raises(RuntimeError, match='simulator crashed'):
    driver(agent.policy, steps=40)
  driver.close()
  assert seen[:8] == [0.0] * 4 + [1.0] * 4


class _Flaky(dummy.Dummy):
  def step(self, action):
    obs = super().step(action)
    obs['log/steps'] = np.float32(self.count)
    if self.count == 3:
      raise ValueError('simulator crashed')
    return obs


def test_worker_extras_and_errors_in_shared_memory_mode():
  """'log/*' keys are not in the observation slab and still arrive; an
  exception inside an env process surfaces in the parent (driver.py:89-99)."""
  agent = make_agent()
  driver = emb.Driver([lambda: _Flaky('disc', length=10)] * 2, parallel=True)
  assert driver._fast
  driver.reset(agent.init_policy)
  seen = []
  driver.on_step(lambda tran, w: seen.append(float(tran['log/steps'])))
  with pytest.raises(RuntimeError, match='simulator crashed'):
    driver(agent.policy, steps=20)
  driver.close()
  assert seen[:6] == [0.0, 0.0, 1.0, 1.0, 2.0, 2.0]
