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


def make_env(length=10):
  return dummy.Dummy('disc', length=length)


def make_agent():
  env = make_env()
  agent = emb.RandomAgent(env.obs_space, env.act_space)
  env.close()
  return agent


@pytest.mark.parametrize('parallel', [False, True])
def test_episode_length(parallel):
  agent = make_agent()
  driver = emb.Driver([make_env], parallel=parallel)
  driver.reset(agent.init_policy)
  seq = []
  driver.on_step(lambda tran, _: seq.append(tran))
  driver(agent.policy, episodes=1)
  driver.close()
  assert len(seq) == 11


def test_first_and_last_step_flags():
  agent = make_agent()
  driver = emb.Driver([make_env], parallel=False)
  driver.reset(agent.init_policy)
  seq = []
  driver.on_step(lambda tran, _: seq.append(tran))
  driver(agent.policy, episodes=2)
  assert len(seq) == 22
  for index in [0, 11]:
    assert seq[index]['is_first'].item() is True
    assert seq[index]['is_last'].item() is False
  for index in [1, 10, 12]:
    assert seq[index]['is_first'].item() is False
  for index in [10, 21]:
    assert seq[index]['is_last'].item() is True
    assert seq[index]['is_first'].item() is False
  for index in [0, 1, 9, 11, 20]:
    assert seq[index]['is_last'].item() is False


def test_env_reset_zeroes_action_and_requests_reset():
  received = []

  class Spy(dummy.Dummy):
    def step(self, action):
      received.append({k: np.array(v) for k, v in action.items()})
      return super().step(action)

  driver = emb.Driver([lambda: Spy('disc', length=5)], parallel=False)
  driver.reset(lambda n: ())
  seq = []
  driver.on_step(lambda tran, _: seq.append(tran))
  action = {'act_disc': np.ones(1, np.int32), 'act_cont': np.zeros((1, 6), np.float32)}
  driver(lambda carry, obs: (carry, action, {}), episodes=2)
  assert len(seq) == 12
  cols = {k: np.array([s[k] for s in seq]) for k in seq[0]}
  assert (cols['is_first'] == [1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0]).all()
  assert (cols['is_last'] == [0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1]).all()
  assert (cols['act_disc'] == [1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0]).all()
  assert 'reset' not in cols
  resets = [bool(a['reset']) for a in received]
  assert resets == [True, False, False, False, False, False, True] + [False] * 5


def test_agent_inputs_and_carry_threading():
  agent = make_agent()
  driver = emb.Driver([make_env], parallel=False)
  driver.reset(agent.init_policy)
  inputs, states = [], []

  def policy(carry, obs, mode='train'):
    inputs.append(obs)
    states.append(carry)
    _, act, _ = agent.policy(carry, obs, mode)
    return 'carry', act, {}

  seq = []
  driver.on_step(lambda tran, _: seq.append(tran))
  driver(policy, episodes=2)
  assert len(seq) == 22
  assert states == ([()] + ['carry'] * 21)
  for index in [0, 11]:
    assert inputs[index]['is_first'].item() is True
  for index in [1, 10, 12, 21]:
    assert inputs[index]['is_first'].item() is False
  for index in [10, 21]:
    assert inputs[index]['is_last'].item() is True


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
  """Driver(envs_per_worker=K): worker w steps envs [w*K, (w+1)*K) one after the
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
