"""Single-process actor + learner loop, the primary caller of the hot path
(reference: embodied/run/train.py:10-118; same signature and `args` fields).

The Driver steps the envs and feeds `replay.add`; the learner runs inside a
Driver callback: `Ratio(train_ratio / (batch_size*batch_length))` train steps
per env step, each `next(stream_train)` -> `agent.train` -> optional
`replay.update(outs['replay'])`.
"""
import collections
import pathlib
from functools import partial as bind

import numpy as np

from .. import utils
from ..core.driver import Driver


def _scalar(x):
  if hasattr(x, 'detach'):
    x = x.detach().cpu().numpy()
  return np.asarray(x)


def train(make_agent, make_replay, make_env, make_stream, make_logger, args):
  agent = make_agent()
  replay = make_replay()
  logger = make_logger()

  logdir = pathlib.Path(args.logdir)
  step = logger.step
  train_agg = utils.Agg()
  epstats = utils.Agg()
  episodes = collections.defaultdict(utils.Agg)
  policy_fps = utils.FPS()
  train_fps = utils.FPS()

  batch_steps = args.batch_size * args.batch_length
  should_train = utils.Ratio(args.train_ratio / batch_steps)
  should_log = utils.LocalClock(args.log_every)
  should_report = utils.LocalClock(args.report_every)
  should_save = utils.LocalClock(args.save_every)

  def logfn(tran, worker):
    # Per-episode statistics (run/train.py:31-54).
    episode = episodes[worker]
    reward = _scalar(tran['reward'])
    if bool(_scalar(tran['is_first'])):
      episode.reset()
    episode.add('score', reward, agg='sum')
    episode.add('length', 1, agg='sum')
    episode.add('rewards', reward, agg='stack')
    for key, value in tran.items():
      if key.startswith('log/'):
        value = _scalar(value)
        assert value.ndim == 0, (key, value.shape, value.dtype)
        episode.add(key, value, agg=('avg', 'max', 'sum'))
    if bool(_scalar(tran['is_last'])):
      result = episode.result()
      logger.add({'score': result.pop('score'), 'length': result.pop('length')},
                 prefix='episode')
      rew = result.pop('rewards')
      if len(rew) > 1:
        result['reward_rate'] = (np.abs(rew[1:] - rew[:-1]) >= 0.01).mean()
      epstats.add(result)

  fns = [bind(make_env, i) for i in range(args.envs)]
  driver = Driver(
      fns, parallel=not args.debug, device=getattr(args, 'device', None))
  driver.on_step(lambda tran, _: step.increment())
  driver.on_step(lambda tran, _: policy_fps.step())
  driver.on_step(replay.add)
  driver.on_step(logfn)

  stream_train = iter(agent.stream(make_stream(replay, 'train')))
  stream_report = iter(agent.stream(make_stream(replay, 'report')))

  carry_train = [agent.init_train(args.batch_size)]
  carry_report = agent.init_report(args.batch_size)

  def trainfn(tran, worker):
    if len(replay) < args.batch_size * args.batch_length:
      return
    for _ in range(should_train(step)):
      with utils.timer.section('stream_next'):
        batch = next(stream_train)
      carry_train[0], outs, mets = agent.train(carry_train[0], batch)
      train_fps.step(batch_steps)
      if 'replay' in outs:
        replay.update(outs['replay'])
      train_agg.add(mets, prefix='train')
  driver.on_step(trainfn)

  cp = utils.Checkpoint(logdir / 'checkpoint.pkl')
  cp.step = step
  cp.agent = agent
  cp.replay = replay
  if getattr(args, 'from_checkpoint', ''):
    import pickle
    data = pickle.loads(pathlib.Path(args.from_checkpoint).read_bytes())
    agent.load(data['agent'])
  cp.load_or_save()

  print('Start training loop')
  policy = lambda *a, **kw: agent.policy(*a, mode='train', **kw)
  driver.reset(agent.init_policy)
  while step < args.steps:

    driver(policy, steps=10)

    if should_report(step) and len(replay):
      agg = utils.Agg()
      for _ in range(args.consec_report * args.report_batches):
        carry_report, mets = agent.report(carry_report, next(stream_report))
        agg.add(mets)
      logger.add(agg.result(), prefix='report')

    if should_log(step):
      logger.add(train_agg.result())
      logger.add(epstats.result(), prefix='epstats')
      logger.add(replay.stats(), prefix='replay')
      logger.add({'fps/policy': policy_fps.result()})
      logger.add({'fps/train': train_fps.result()})
      logger.add({'timer': utils.timer.stats()['summary']})
      logger.write()

    if should_save(step):
      cp.save()

  logger.close()
  driver.close()
