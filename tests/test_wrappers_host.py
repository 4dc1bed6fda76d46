"""Env wrappers (host glue around single envs) against golden vectors recorded
from the reference's wrappers.py through its own `wrap_env` chain.  CPU only."""
import numpy as np
import pytest

from tests import adapters, scenarios
from tests.conftest import assert_same, load_golden


@pytest.mark.parametrize('name', sorted(scenarios.HOST_SCENARIOS))
def test_wrappers_match_reference_golden(name):
  got = scenarios.HOST_SCENARIOS[name](adapters.product_ns())
  assert_same(got, load_golden(name), name)


@pytest.mark.reference
@pytest.mark.parametrize('name', sorted(scenarios.HOST_SCENARIOS))
def test_host_golden_is_current(name):
  from oracle import refload
  if not refload.available():
    pytest.skip('no /root/reference here')
  got = scenarios.HOST_SCENARIOS[name](adapters.reference_ns())
  assert_same(got, load_golden(name), name)


def test_wrapped_envs_run_under_the_driver():
  """The wrap_env chain in front of the host Driver: dtypes that reach the
  policy are the unified ones."""
  import embodied_amd as emb
  from embodied_amd.core import wrappers as W

  def make():
    env = scenarios._ScriptedEnv(emb.Space)
    env = W.NormalizeAction(env, 'action')
    env = W.UnifyDtypes(env)
    env = W.CheckSpaces(env)
    env = W.ClipAction(env, 'action')
    return W.TimeLimit(env, 5)

  driver = emb.Driver([make, make], parallel=False)
  seen = []
  driver.on_step(lambda tran, worker, **kw: seen.append(tran))

  def policy(carry, obs):
    n = len(obs['is_first'])
    assert obs['vec'].dtype == np.float32 and obs['count'].dtype == np.int32
    act = {'action': np.full((n, 2), 3.0, np.float32), 'free': np.zeros((n, 2), np.float32),
           'choice': np.ones(n, np.int32)}
    return carry, act, {}

  driver.reset()
  driver(policy, steps=24)
  assert len(seen) == 24
  assert any(t['is_last'] for t in seen)
