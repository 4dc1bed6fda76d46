"""Policy-only loop (reference: embodied/run/eval_only.py:9-74, same
signature): steps the envs with `mode='eval'`, logs episode statistics."""
from functools import partial as bind

from .. import utils
from ..core.driver import Driver
from .stats import EpisodeStats


def eval_only(make_agent, make_env, make_logger, args):
  agent, logger = make_agent(), make_logger()
  step = logger.step
  epstats = utils.Agg()
  policy_fps = utils.FPS()
  should_log = utils.LocalClock(args.log_every)
  driver = Driver(
      [bind(make_env, index) for index in range(args.envs)],
      parallel=not args.debug, device=getattr(args, 'device', None))
  episodes = EpisodeStats(logger, epstats)

  def count(trans, workers, **kw):
    step.increment(args.envs)
    policy_fps.step(args.envs)

  driver.on_batch(count)
  driver.on_batch(episodes.on_batch)
  if getattr(args, 'from_checkpoint', ''):
    checkpoint = utils.Checkpoint()
    checkpoint.agent = agent
    checkpoint.load(args.from_checkpoint, keys=['agent'])

  print('Start evaluation')
  policy = lambda *a, **kw: agent.policy(*a, mode='eval', **kw)
  driver.reset(agent.init_policy)
  while step < args.steps:
    driver(policy, steps=10)
    if should_log(step):
      logger.add(epstats.result(), prefix='epstats')
      logger.add({'fps/policy': policy_fps.result()})
      logger.write()
  episodes.flush()
  logger.close()
  driver.close()
