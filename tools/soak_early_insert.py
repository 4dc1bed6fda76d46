"""Soak of the stepping loop's concurrency: the device Driver steps through the
early insert (index bookkeeping on the library's helper thread) while sampler
threads draw batches on streams of their own (actor / learner layout:
emb_replay_multistream orders pool writes and reads).  Every sampled window is
checked against the synthetic env's generator: frame byte i of a step is
(salt + i) & 0xFF, the salt advances by 7 per step and restarts where is_first
is set, the reward is the episode step modulo 7, the stored action is the
policy's tick, zeroed where the episode ended.
  python tools/soak_early_insert.py --seconds 60"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embodied_amd as emb
from embodied_amd.envs import synthetic

p = argparse.ArgumentParser()
p.add_argument('--seconds', type=float, default=30)
p.add_argument('--samplers', type=int, default=2)
p.add_argument('--envs', type=int, default=64)
p.add_argument('--unmasked', action='store_true',
               help='the env takes unmasked actions: the action write is carried into the next early insert')
args = p.parse_args()

n, L = args.envs, 16
env = synthetic.SyntheticBatchEnv(n, episode_len=37, ring=4, takes_unmasked_actions=args.unmasked)
rep = emb.Replay(length=L, capacity=20000, chunksize=256, seed=0)
driver = emb.Driver(batch_env=env, device='cuda')
driver.on_step(rep.add)
stage = torch.empty((n, 4, 84, 84), dtype=torch.bfloat16, device='cuda')
ticks = torch.arange(1, 1 << 16, dtype=torch.int32, device='cuda')
tick = [0]


def policy(carry, obs, **kw):
  emb.ops.obs_stack(obs['image'], layout='channels_first', dtype=torch.bfloat16, scale=1 / 255, out=stage)
  t = tick[0] = (tick[0] + 1) % (len(ticks) - 1)
  return carry, {'action': ticks[t].expand(n)}, {}


errors, counts, running = [], {'steps': 0, 'windows': 0}, [True]


def check(batch):
  img = batch['image'].flatten(2)                    # (B, L, 28224) uint8
  salt = img[:, :, 0].to(torch.int32)
  first = batch['is_first']
  last = batch['is_last']
  bad = []
  if not torch.equal(img[:, :, -1].to(torch.int32), (salt + img.shape[2] - 1) & 0xFF):
    bad.append('frame tail')
  if not torch.equal(img[:, :, 4097].to(torch.int32), (salt + 4097) & 0xFF):
    bad.append('frame middle')
  follows = ((salt[:, 1:] - salt[:, :-1]) & 0xFF) == 7
  if not bool((follows | first[:, 1:]).all()):
    bad.append('salt sequence')
  if not bool((first[:, 1:] == last[:, :-1]).all()):
    bad.append('is_first after is_last')
  act = batch['action']
  step_ok = (act[:, 1:] - act[:, :-1] == 1) | last[:, 1:] | last[:, :-1] | (act[:, :-1] >= len(ticks) - 2)
  if not bool(step_ok.all()) or not bool((act[last] == 0).all()):
    bad.append('actions')
  return bad


def sampler(k):
  stream = torch.cuda.Stream()
  while running[0]:
    if len(rep) < 64:
      time.sleep(0.01)
      continue
    try:
      with torch.cuda.stream(stream):
        batch = rep.sample(8)
        bad = check(batch)
      if bad:
        errors.append((k, bad))
      counts['windows'] += 8
    except Exception as e:
      errors.append((k, repr(e)))
      return


threads = [threading.Thread(target=sampler, args=(k,), daemon=True) for k in range(args.samplers)]
driver.reset()
driver(policy, steps=n * 32)
[t.start() for t in threads]
end = time.time() + args.seconds
while time.time() < end and not errors:
  driver(policy, steps=n * 200)
  counts['steps'] += n * 200
running[0] = False
[t.join() for t in threads]
torch.cuda.synchronize()
print('env steps', counts['steps'], 'windows checked', counts['windows'], 'early inserts', rep.early_inserts,
      'deferred', rep.profile_report('deferred')[0], 'carried (inline, all)', rep.profile_report('carried')[:2],
      'errors', errors[:5])
sys.exit(1 if errors else 0)
