"""BASELINE configs[4] end to end: Director on a dmc_humanoid-shaped task.

The reference's Director keeps ONE flat Replay(batch_length, replay_size)
(director/train.py:133-144; batch 16 x 64, director/configs.yaml:77-78); its two
levels are reshapes of imagined trajectories of H+1 = 16 steps started from
every one of the 16*64 = 1024 sampled states (director/hierarchy.py:224-256,
train_skill_duration k = 8, imag_horizon 15) and the critic target is
`VFunction.score` over them (director/agent.py:430-445).

One test chains the whole path on the GPU — device Driver -> flat Replay ->
sample (16, 64) -> time-major trajectories (16, 1024) -> split_traj /
abstract_traj -> score for the worker (k, 2*1024) and manager (2, 1024) views —
and checks every stage against the CPU oracle, which is itself pinned to the
reference's source by tests/test_scan_golden.py.
"""
import numpy as np
import pytest
import torch

from oracle import np_oracle
from tests.conftest import assert_same

pytestmark = pytest.mark.gpu

B, T, H, K = 16, 64, 15, 8          # batch, batch_length, imag_horizon, skill duration
N_ENVS, ACT = 8, 21                 # humanoid: 21 actuators; 64x64x3 vision frames
TOL = dict(rtol=1e-5, atol=1e-5)


class HumanoidLikeEnv:
  """Host env with dmc_vision humanoid shapes and scripted content."""

  def __init__(self, index, length=37):
    self.index, self.length, self.t = index, length + index, 0
    self.gen = np.random.default_rng(index)

  @property
  def obs_space(self):
    import embodied_amd as emb
    return {'image': emb.Space(np.uint8, (64, 64, 3)), 'reward': emb.Space(np.float32),
            'is_first': emb.Space(bool), 'is_last': emb.Space(bool), 'is_terminal': emb.Space(bool)}

  @property
  def act_space(self):
    import embodied_amd as emb
    return {'action': emb.Space(np.float32, (ACT,), -1, 1), 'reset': emb.Space(bool)}

  def step(self, action):
    if action['reset'] or self.t >= self.length:
      self.t = 0
    else:
      self.t += 1
    done = self.t >= self.length
    return {
        'image': self.gen.integers(0, 255, (64, 64, 3), dtype=np.uint8),
        'reward': np.float32(np.sin(0.1 * self.t + self.index) + float(action['action'][0])),
        'is_first': self.t == 0, 'is_last': done, 'is_terminal': done and self.index % 2 == 0}

  def close(self):
    pass


def test_director_config_end_to_end():
  import embodied_amd as emb
  capacity = 600
  ours = emb.Replay(length=T, capacity=capacity, chunksize=128, seed=11)
  ref = np_oracle.Replay(T, capacity, 128, seed=11)
  gen = np.random.default_rng(5)
  script = gen.uniform(-1, 1, (400, N_ENVS, ACT)).astype(np.float32)
  tick = {'ours': 0, 'ref': 0}

  def policy_for(name):
    def policy(carry, obs, **kw):
      act = script[tick[name] % len(script)]
      tick[name] += 1
      return carry, {'action': act}, {}
    return policy

  # --- Driver -> flat Replay (product on the GPU, oracle on the host, same envs) ---
  driver = emb.Driver([lambda i=i: HumanoidLikeEnv(i) for i in range(N_ENVS)],
                      parallel=False, device='cuda')
  driver.on_step(ours.add)
  oracle_driver = np_oracle.Driver([HumanoidLikeEnv(i) for i in range(N_ENVS)])
  oracle_driver.on_step(ref.add)
  driver(policy_for('ours'), steps=N_ENVS * 150)
  for _ in range(150):
    oracle_driver.step(policy_for('ref'))
  assert len(ours) == len(ref) == capacity

  # --- sample (B, T): bit-exact ---
  got = ours.sample(B)
  want = ref.sample(B)
  assert got['image'].shape == (B, T, 64, 64, 3) and got['action'].shape == (B, T, ACT)
  assert_same({k: v.cpu().numpy() for k, v in got.items()}, want, 'director sample')

  # --- imagined trajectories: (H+1, B*T), every sampled state is a start ---
  n = B * T
  flat = lambda x: x.reshape(n, *x.shape[2:])
  rng = np.random.default_rng(9)
  start_rew = flat(want['reward'])                                  # (1024,)
  decay = np.float32(0.9) ** np.arange(H, dtype=np.float32)[:, None]
  traj = {
      'reward_extr': (start_rew[None] * decay + rng.standard_normal((H, n)) * 0.1).astype(np.float32),
      'reward_goal': rng.standard_normal((H, n)).astype(np.float32),
      'cont': np.where(rng.random((H + 1, n)) < 0.03, 0.0,
                       1.0 - 0.5 * flat(want['is_terminal'])[None]).astype(np.float32),
      'value': rng.standard_normal((H + 1, n)).astype(np.float32),
      'action': np.broadcast_to(flat(want['action'])[None], (H + 1, n, ACT)).copy(),
  }
  dev = {k: torch.as_tensor(v).cuda() for k, v in traj.items()}

  # --- worker windows (split_traj) and manager steps (abstract_traj), k = 8 ---
  worker, manager = {}, {}
  for key in traj:
    reward = key.startswith('reward_')
    kind = 'reward' if reward else 'cont' if key == 'cont' else 'first'
    worker[key] = emb.scans.split_traj(dev[key], K, reward)
    manager[key] = emb.scans.abstract_traj(dev[key], dev['cont'], K, kind)
    assert np.array_equal(worker[key].cpu().numpy(), np_oracle.split_traj(traj[key], K, reward)), key
    np.testing.assert_allclose(manager[key].cpu().numpy(),
                               np_oracle.abstract_traj(traj[key], traj['cont'], K, kind), **TOL)
  assert worker['cont'].shape == (K, 2 * n) and worker['reward_extr'].shape == (K - 1, 2 * n)
  assert manager['cont'].shape == (2, n) and manager['reward_extr'].shape == (1, n)

  # --- critic targets: full trajectory (16, 1024), worker (8, 2048), manager (2, 1024) ---
  views = {
      'flat': (dev['reward_extr'], dev['cont'], dev['value']),
      'worker': (worker['reward_goal'].contiguous(), worker['cont'].contiguous(), worker['value'].contiguous()),
      'manager': (manager['reward_extr'], manager['cont'], manager['value'].contiguous()),
  }
  for name, (rew, cont, value) in views.items():
    ret = emb.scans.director_score(rew, cont, value, horizon=333, lam=0.95)
    want_ret = np_oracle.director_score(
        rew.cpu().numpy(), cont.cpu().numpy(), value.cpu().numpy(), 333, 0.95)
    assert ret.shape == rew.shape, name
    np.testing.assert_allclose(ret.cpu().numpy(), want_ret, err_msg=name, **TOL)
