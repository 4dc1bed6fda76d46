"""Actor / learner split in ONE process on ONE GPU: an actor thread steps the
Driver on its own HIP stream and inserts into the Replay, a learner thread
samples, trains and writes back on a second stream, coupled by the
samples-per-insert limiter.

This is the reference's `parallel_actor` / `parallel_learner` /
`parallel_replay` triangle (embodied/run/parallel.py:62-314) with the three TCP
hops replaced by shared device memory: same limiter settings
(`parallel.py:242-245`), same rule that every inserted step takes an insert
token and every sampled sequence a sample token (`:247-275`), and the learner
applies `replay.update(outs['replay'])` one train step late, which is how the
reference's agent hands out its outputs (embodied/jax/agent.py:286-294).
The Replay orders pool writes and reads across the two streams with events
(`emb_replay_multistream`).
"""
import threading
import time
from functools import partial as bind

import torch

from .. import utils
from ..core import limiters
from ..core.driver import Driver
from .stats import EpisodeStats


def actor_learner(make_agent, make_replay, make_env, make_stream, make_logger, args):
  agent, replay, logger = make_agent(), make_replay(), make_logger()
  step = logger.step
  device = torch.device(getattr(args, 'device', 'cuda'))
  epstats = utils.Agg()
  train_agg = utils.Agg()
  policy_fps, train_fps = utils.FPS(), utils.FPS()
  should_log = utils.LocalClock(args.log_every)
  batch_steps = args.batch_size * args.batch_length
  limiter = limiters.SamplesPerInsert(
      args.train_ratio / args.batch_length,
      tolerance=4 * args.batch_size,
      minsize=args.batch_size * replay.length)
  running = [True]
  errors = []
  counters = {'trains': 0, 'insert_waits': 0, 'sample_waits': 0}

  driver = Driver(
      [bind(make_env, i) for i in range(args.envs)], parallel=not args.debug,
      device=device)
  episodes = EpisodeStats(logger, epstats)
  n = args.envs

  def gate_and_count(trans, workers, **kw):
    # One insert token per env step, waiting while the learner lags.
    for _ in range(n):
      if limiters.wait(lambda: limiter.want_insert() or not running[0],
                       'Replay insert waiting', sleep=0.0005):
        counters['insert_waits'] += 1
      limiter.insert()
    step.increment(n)
    policy_fps.step(n)

  driver.on_batch(gate_and_count)
  driver.on_step(replay.add)
  driver.on_batch(episodes.on_batch)

  def actor():
    try:
      with torch.cuda.stream(torch.cuda.Stream(device)):
        policy = lambda *a, **kw: agent.policy(*a, mode='train', **kw)
        driver.reset(agent.init_policy)
        while running[0] and step < args.steps:
          driver(policy, steps=n)
    except BaseException as e:
      errors.append(e)
    finally:
      running[0] = False

  def learner():
    try:
      with torch.cuda.stream(torch.cuda.Stream(device)):
        stream = iter(agent.stream(make_stream(replay, 'train')))
        carry = agent.init_train(args.batch_size)
        late = None
        while running[0]:
          for _ in range(args.batch_size):
            if limiters.wait(lambda: limiter.want_sample() or not running[0],
                             'Replay sample waiting', sleep=0.0005):
              counters['sample_waits'] += 1
            if not running[0]:
              return
            limiter.sample()
          batch = next(stream)
          carry, outs, mets = agent.train(carry, batch)
          if late is not None:
            replay.update(late)          # previous step's outputs: one step late
          late = outs.get('replay')
          counters['trains'] += 1
          train_fps.step(batch_steps)
          train_agg.add(mets, prefix='train')
    except BaseException as e:
      errors.append(e)
      running[0] = False

  threads = [threading.Thread(target=actor, name='actor'),
             threading.Thread(target=learner, name='learner')]
  [t.start() for t in threads]
  try:
    while running[0]:
      time.sleep(0.02)
      if should_log(step):
        logger.add(train_agg.result())
        logger.add(epstats.result(), prefix='epstats')
        logger.add(replay.stats(), prefix='replay')
        logger.add({'fps/policy': policy_fps.result(), 'fps/train': train_fps.result(),
                    'limiter/avail': limiter.avail, **{f'limiter/{k}': v for k, v in counters.items()}})
        logger.write()
  finally:
    running[0] = False
    [t.join() for t in threads]
    episodes.flush()
    driver.close()
    logger.add({f'limiter/{k}': v for k, v in counters.items()})
    logger.close()
  if errors:
    raise errors[0]
  return counters
