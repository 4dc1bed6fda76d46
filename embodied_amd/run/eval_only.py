"""Policy-only loop (reference: embodied/run/eval_only.py:9-74)."""
import collections
from functools import partial as bind

import numpy as np

from .. import utils
from ..core.driver import Driver
from .train import _scalar


def eval_only(make_agent, make_env, make_logger, args):
  agent = make_agent()
  logger = make_logger()
  step = logger.step
  episodes = collections.defaultdict(utils.Agg)
  epstats = utils.Agg()
  policy_fps = utils.FPS()
  should_log = utils.LocalClock(args.log_every)

  def logfn(tran, worker):
    episode = episodes[worker]
    if bool(_scalar(tran['is_first'])):
      episode.reset()
    episode.add('score', _scalar(tran['reward']), agg='sum')
    episode.add('length', 1, agg='sum')
    if bool(_scalar(tran['is_last'])):
      result = episode.result()
      logger.add({'score': result.pop('score'), 'length': result.pop('length')},
                 prefix='episode')
      epstats.add(result)

  fns = [bind(make_env, i) for i in range(args.envs)]
  driver = Driver(fns, parallel=not args.debug, device=getattr(args, 'device', None))
  driver.on_step(lambda tran, _: step.increment())
  driver.on_step(lambda tran, _: policy_fps.step())
  driver.on_step(logfn)

  if getattr(args, 'from_checkpoint', ''):
    cp = utils.Checkpoint()
    cp.agent = agent
    cp.load(args.from_checkpoint, keys=['agent'])

  print('Start evaluation')
  policy = lambda *a, **kw: agent.policy(*a, mode='eval', **kw)
  driver.reset(agent.init_policy)
  while step < args.steps:
    driver(policy, steps=10)
    if should_log(step):
      logger.add(epstats.result(), prefix='epstats')
      logger.add({'fps/policy': policy_fps.result()})
      logger.write()
  logger.close()
  driver.close()
