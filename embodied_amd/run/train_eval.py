"""Training with a second, evaluation Driver/Replay pair
(reference: embodied/run/train_eval.py:10-157; same signature)."""
import pathlib
from functools import partial as bind

from .. import utils
from ..core.driver import Driver
from .train import _Learner


def train_eval(
    make_agent, make_replay_train, make_replay_eval, make_env_train,
    make_env_eval, make_stream, make_logger, args):
  agent = make_agent()
  replay_train = make_replay_train()
  replay_eval = make_replay_eval()
  logger = make_logger()
  step = logger.step
  logdir = pathlib.Path(args.logdir)
  policy_fps = utils.FPS()
  should_log = utils.LocalClock(args.log_every)
  should_eval = utils.LocalClock(getattr(args, 'eval_every', args.report_every), first=True)
  should_save = utils.LocalClock(args.save_every)
  device = getattr(args, 'device', None)

  fns = [bind(make_env_train, i) for i in range(args.envs)]
  driver_train = Driver(fns, parallel=not args.debug, device=device)
  driver_train.on_batch(lambda trans, workers, **kw: (step.increment(args.envs), policy_fps.step(args.envs)))
  driver_train.on_step(replay_train.add)

  fns = [bind(make_env_eval, i) for i in range(getattr(args, 'eval_envs', 1))]
  driver_eval = Driver(fns, parallel=not args.debug, device=device)
  driver_eval.on_step(replay_eval.add)

  stream_train = iter(agent.stream(make_stream(replay_train, 'train')))
  stream_report = iter(agent.stream(make_stream(replay_train, 'report')))
  stream_eval = iter(agent.stream(make_stream(replay_eval, 'eval')))
  carry_report = agent.init_report(args.batch_size)
  carry_eval = agent.init_report(args.batch_size)

  learner = _Learner(agent, replay_train, stream_train, step, args)
  driver_train.on_batch(learner)

  cp = utils.Checkpoint(logdir / 'checkpoint.pkl')
  cp.step = step
  cp.agent = agent
  cp.replay_train = replay_train
  cp.replay_eval = replay_eval
  cp.load_or_save()

  train_policy = lambda *a, **kw: agent.policy(*a, mode='train', **kw)
  eval_policy = lambda *a, **kw: agent.policy(*a, mode='eval', **kw)
  driver_train.reset(agent.init_policy)
  while step < args.steps:
    if should_eval(step):
      driver_eval.reset(agent.init_policy)
      driver_eval(eval_policy, episodes=getattr(args, 'eval_eps', 1))
      if len(replay_eval):
        carry_eval, mets = agent.report(carry_eval, next(stream_eval))
        logger.add(mets, prefix='eval')
      if len(replay_train):
        carry_report, mets = agent.report(carry_report, next(stream_report))
        logger.add(mets, prefix='report')
    driver_train(train_policy, steps=10)
    if should_log(step):
      logger.add(learner.metrics.result())
      logger.add(replay_train.stats(), prefix='replay')
      logger.add({'fps/policy': policy_fps.result(), 'fps/train': learner.fps.result()})
      logger.write()
    if should_save(step):
      cp.save()
  logger.close()
  driver_train.close()
  driver_eval.close()
