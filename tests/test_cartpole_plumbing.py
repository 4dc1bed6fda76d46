"""BASELINE config 1: CartPole, 4 parallel envs, numpy Driver on the CPU with a
random agent — pure plumbing, no GPU."""
import numpy as np

import embodied_amd as emb
from embodied_amd.envs import cartpole


def test_four_parallel_cartpoles_with_random_agent():
  env = cartpole.CartPole()
  act_space = {k: v for k, v in env.act_space.items() if k != 'reset'}
  agent = emb.RandomAgent(env.obs_space, act_space)
  driver = emb.Driver(
      [(lambda i=i: cartpole.CartPole(seed=i)) for i in range(4)], parallel=True)
  driver.reset(agent.init_policy)
  lengths, current = [], np.zeros(4, int)

  def track(tran, worker):
    current[worker] += 1
    assert tran['vector'].shape == (4,) and tran['vector'].dtype == np.float32
    if tran['is_last']:
      lengths.append(current[worker])
      current[worker] = 0

  driver.on_step(track)
  driver(agent.policy, episodes=12)
  driver.close()
  assert len(lengths) >= 12
  # a random policy balances for ~20 steps on average; never the 500-step limit
  assert 8 <= np.mean(lengths) <= 60


def test_cartpole_dynamics_are_deterministic_per_seed():
  a, b = cartpole.CartPole(seed=3), cartpole.CartPole(seed=3)
  act = {'reset': True, 'action': np.int32(0)}
  for t in range(50):
    oa, ob = a.step(act), b.step(act)
    assert np.array_equal(oa['vector'], ob['vector'])
    act = {'reset': False, 'action': np.int32(t % 2)}
