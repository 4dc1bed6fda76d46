"""A small torch PPO agent written against the `Agent` protocol, to show the
hot path under a real actor-learner: device Driver -> Replay (online) ->
Consec stream -> GAE kernel -> update.  Only enough model to exercise the path
(SURVEY.md 7, step 10): an MLP / small conv encoder, categorical policy, value
head, Adam.  Loss structure follows ppo/agent.py:177-235 (clipped surrogate,
value regression to the GAE target); normalisers are plain batch statistics.

    python examples/ppo_torch.py --steps 60000       # CartPole, 16 envs, GPU
"""
import argparse
import pathlib
import sys
import types

import torch
import torch.nn as nn

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import embodied_amd as emb  # noqa: E402


class PPOAgent(emb.Agent):

  def __init__(self, obs_space, act_space, config):
    self.obs_space, self.act_space, self.config = obs_space, act_space, config
    self.device = torch.device(config.device)
    (self.act_key, space), = [(k, v) for k, v in act_space.items() if k != 'reset']
    self.n_actions = int(space.high.max()) if space.shape == () else None
    self.image = 'image' in obs_space
    if self.image:
      c = obs_space['image'].shape[-1]
      self.enc = nn.Sequential(
          nn.Conv2d(c, 32, 8, 4), nn.ReLU(), nn.Conv2d(32, 64, 4, 2), nn.ReLU(),
          nn.Flatten(), nn.LazyLinear(256), nn.ReLU())
    else:
      self.enc = nn.Sequential(
          nn.Linear(obs_space['vector'].shape[0], 128), nn.Tanh(),
          nn.Linear(128, 128), nn.Tanh())
    feat = 256 if self.image else 128
    self.pi = nn.Linear(feat, self.n_actions)
    self.vf = nn.Linear(feat, 1)
    self.model = nn.ModuleDict(dict(enc=self.enc, pi=self.pi, vf=self.vf)).to(self.device)
    self.opt = None

  def _features(self, obs):
    if self.image:
      frames = obs['image']
      lead = frames.shape[:-3]
      batch = emb.ops.obs_stack(     # (N, H, W, C) u8 -> (N, C, H, W) f32 in [0, 1]
          frames.reshape(-1, *frames.shape[-3:]), layout='channels_first',
          dtype=torch.float32, scale=1 / 255)
      return self.enc(batch).reshape(*lead, -1)
    return self.enc(obs['vector'].to(self.device, torch.float32))

  def init_policy(self, batch_size):
    return ()

  init_train = init_report = init_policy

  @torch.no_grad()
  def policy(self, carry, obs, mode='train'):
    feat = self._features(obs)
    dist = torch.distributions.Categorical(logits=self.pi(feat))
    action = dist.sample() if mode == 'train' else dist.probs.argmax(-1)
    outs = {'logp': dist.log_prob(action), 'value': self.vf(feat)[..., 0]}
    return carry, {self.act_key: action.to(torch.int32)}, outs

  def train(self, carry, data):
    if self.opt is None:
      self.opt = torch.optim.Adam(self.model.parameters(), lr=self.config.lr)
    cfg = self.config
    feat = self._features(data)
    dist = torch.distributions.Categorical(logits=self.pi(feat))
    value = self.vf(feat)[..., 0]
    logp = dist.log_prob(data[self.act_key].long())
    # Return scan on the GPU: one kernel for the whole (B, T) batch.
    adv, tar = emb.scans.gae(
        data['reward'], value.detach(), data['is_last'], data['is_terminal'],
        hor=cfg.horizon, lam=cfg.lam)
    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    mask = (~(data['is_last'] | data['is_terminal']))[:, :-1].float()
    ratio = torch.exp(logp - data['logp'])[:, :-1]
    surr = torch.minimum(ratio * adv, ratio.clamp(1 - cfg.clip, 1 + cfg.clip) * adv)
    loss_pi = -(surr * mask).mean() - cfg.entropy * (dist.entropy()[:, :-1] * mask).mean()
    loss_v = ((value[:, :-1] - tar) ** 2 * mask).mean()
    loss = loss_pi + 0.5 * loss_v
    self.opt.zero_grad(set_to_none=True)
    loss.backward()
    nn.utils.clip_grad_norm_(self.model.parameters(), 10.0)
    self.opt.step()
    mets = {'loss': loss.detach(), 'value_loss': loss_v.detach(), 'adv_mag': adv.abs().mean()}
    return carry, {}, {k: v.cpu().numpy() for k, v in mets.items()}

  def report(self, carry, data):
    return carry, {}

  def stream(self, st):
    return st

  def save(self):
    return {'model': {k: v.cpu() for k, v in self.model.state_dict().items()}}

  def load(self, data):
    self.model.load_state_dict(data['model'])


def main(argv=None):
  p = argparse.ArgumentParser()
  p.add_argument('--steps', type=int, default=60000)
  p.add_argument('--envs', type=int, default=16)
  p.add_argument('--logdir', default='/tmp/ppo_cartpole')
  flags = p.parse_args(argv)
  from embodied_amd.envs import cartpole
  cfg = types.SimpleNamespace(device='cuda', lr=3e-4, horizon=200, lam=0.8, clip=0.2, entropy=1e-2)
  args = types.SimpleNamespace(
      logdir=flags.logdir, batch_size=16, batch_length=32, train_ratio=8.0,
      log_every=1, report_every=1e9, save_every=1e9, envs=flags.envs, debug=True,
      from_checkpoint='', steps=flags.steps, consec_report=1, report_batches=1,
      device='cuda')
  env0 = cartpole.CartPole()
  obs_space = {k: v for k, v in env0.obs_space.items()}
  act_space = {k: v for k, v in env0.act_space.items() if k != 'reset'}
  logger = emb.utils.Logger(printer=print)
  emb.run.train(
      lambda: PPOAgent(obs_space, act_space, cfg),
      lambda: emb.Replay(length=args.batch_length + 1, capacity=20000, chunksize=256, online=True),
      lambda i: cartpole.CartPole(seed=i),
      lambda replay, mode: emb.streams.Consec(
          emb.streams.Stateless(replay.sample, args.batch_size, mode),
          length=args.batch_length, consec=1, prefix=1, strict=True, contiguous=True),
      lambda: logger, args)
  return logger


if __name__ == '__main__':
  main()
