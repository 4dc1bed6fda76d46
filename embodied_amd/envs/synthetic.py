"""Device-resident synthetic vector env: N simulators' worth of 84x84x4 uint8
frames generated straight into HBM by one kernel (`emb_synth_env_step`), with
the episode logic of the reference's Dummy env (embodied/envs/dummy.py:38-48).
This is the benchmark/test input of SURVEY.md 8d: real simulators are CPU code
outside the hot path.  `host_step` restates the same generator in numpy so the
frames can be verified and a CPU baseline can run on identical data.
"""
import numpy as np
import torch

from ..space import Space
from .. import _lib
from .._lib import api, fast


class SyntheticBatchEnv:

  def __init__(self, n, shape=(84, 84, 4), episode_len=1000, env0=0,
               actions=6, device='cuda', ring=0, takes_unmasked_actions=False):
    self.n = n
    # True: the Driver may hand this env the policy's actions unmasked, with
    # `reset` beside them (this generator never looks at the action of an env
    # it restarts -- nor at any other): see Driver._step_device_env.
    self.takes_unmasked_actions = bool(takes_unmasked_actions)
    self.shape = tuple(shape)
    self.frame_bytes = int(np.prod(shape))
    assert self.frame_bytes % 16 == 0
    self.episode_len = episode_len
    self.env0 = env0
    self.actions = actions
    self.device = torch.device(device)
    if self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    # {count, done} per env in two generations: a step reads one and writes the
    # other, so every workgroup of an env may read the state while one writes.
    self.counters = torch.zeros((2, 2 * n), dtype=torch.int32, device=self.device)
    self.generation = 0
    # ring > 0: observations are written into `ring` rotating sets of output
    # buffers instead of fresh tensors (like vector envs that own their output
    # arrays): a returned dict is overwritten `ring` steps later.
    self.ring = [self._alloc() for _ in range(ring)]
    # (the ring's tensors never change: their addresses are taken once)
    self._ring_ptrs = [tuple(v.data_ptr() for v in obs.values()) for obs in self.ring]
    self._counters_ptr = self.counters.data_ptr()
    self._calls = {}
    self.turn = 0

  def __len__(self):
    return self.n

  @property
  def obs_space(self):
    return {
        'image': Space(np.uint8, self.shape),
        'reward': Space(np.float32),
        'is_first': Space(bool),
        'is_last': Space(bool),
        'is_terminal': Space(bool),
    }

  @property
  def act_space(self):
    return {
        'reset': Space(bool),
        'action': Space(np.int32, (), 0, self.actions),
    }

  def _alloc(self):
    n, dev = self.n, self.device
    return {
        'image': _lib.empty((n, *self.shape), torch.uint8, dev),
        'reward': _lib.empty((n,), torch.float32, dev),
        'is_first': _lib.empty((n,), torch.bool, dev),
        'is_last': _lib.empty((n,), torch.bool, dev),
        'is_terminal': _lib.empty((n,), torch.bool, dev),
    }

  def step(self, acts):
    n, dev = self.n, self.device
    reset = acts['reset']
    if self.ring:
      turn = self.turn
      obs = dict(self.ring[turn])
      self.turn = (turn + 1) % len(self.ring)
      # The Driver passes the previous step's is_last as `reset`: with an output
      # ring the launch's arguments repeat with the ring (and the counters'
      # generation) -- remembered per (turn, generation) while `reset` is the
      # same tensor object.
      cached = self._calls.get((turn, self.generation))
      if cached is not None and cached[0] is reset:
        fast.emb_synth_env_step(*cached[1], _lib.raw_stream(dev))
        self.generation ^= 1
        return obs
      image, reward, is_first, is_last, is_terminal = self._ring_ptrs[turn]
    else:
      turn = -1
      obs = self._alloc()
      image, reward, is_first, is_last, is_terminal = (v.data_ptr() for v in obs.values())
    reset_ptr = reset.data_ptr()
    if reset_ptr == is_last or reset_ptr == is_first or reset_ptr == is_terminal:
      # `reset` is a flag buffer this very step writes (ring=1: the Driver passes
      # the previous is_last): the workgroups that share an env's frame would
      # read it before and after workgroup 0's store -- step on a copy.
      reset = reset.clone()
      reset_ptr = reset.data_ptr()
      turn = -1                   # (a fresh copy per step: nothing to remember)
    args = (image, reward, is_first, is_last, is_terminal, n, self.frame_bytes, self.env0,
            self.episode_len, reset_ptr, self._counters_ptr, self.generation)
    fast.emb_synth_env_step(*args, _lib.raw_stream(dev))
    if turn >= 0:
      self._calls[(turn, self.generation)] = (reset, args)
    self.generation ^= 1
    return obs


_RAMPS = {}


def host_frame(env, count, frame_bytes):
  """Byte i of env `env`'s frame at episode step `count`:
  (env*131 + count*7 + i) & 0xFF -- a uint8 ramp plus a uint8 salt (uint8
  addition wraps modulo 256)."""
  ramp = _RAMPS.get(frame_bytes)
  if ramp is None:
    ramp = _RAMPS[frame_bytes] = (np.arange(frame_bytes, dtype=np.int64) & 0xFF).astype(np.uint8)
  salt = np.uint8((env * 131 + count * 7) & 0xFF)
  return ramp + salt


class HostSyntheticEnv:
  """One env of the same generator on the host (numpy), `Env` protocol."""

  def __init__(self, env, shape=(84, 84, 4), episode_len=1000, actions=6):
    self.env = env
    self.shape = tuple(shape)
    self.frame_bytes = int(np.prod(shape))
    self.length = episode_len + (env % 8) * 13
    self.actions = actions
    self.count = 0
    self.done = False

  obs_space = property(lambda self: SyntheticBatchEnv.obs_space.fget(self))
  act_space = property(lambda self: SyntheticBatchEnv.act_space.fget(self))

  def step(self, action):
    restart = bool(action['reset']) or self.done
    if restart:
      self.count, self.done = 0, False
    else:
      self.count += 1
      self.done = self.count >= self.length
    return {
        'image': host_frame(self.env, self.count, self.frame_bytes).reshape(self.shape),
        'reward': np.float32(0 if restart else self.count % 7),
        'is_first': restart,
        'is_last': self.done,
        'is_terminal': self.done,
    }

  def close(self):
    pass
