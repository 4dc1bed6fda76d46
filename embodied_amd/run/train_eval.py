"""Training with a second, evaluation Driver/Replay pair.

Same entry point and `args` fields as the reference's
`embodied.run.train_eval(make_agent, make_replay_train, make_replay_eval,
make_env_train, make_env_eval, make_stream, make_logger, args)`
(embodied/run/train_eval.py:10-157).  Two halves share one agent:

* the training half is `run.train`'s: `args.envs` envs feed the train Replay,
  the learner runs as a Driver callback;
* the evaluation half (`_Evaluation`) owns `args.eval_envs` envs and the eval
  Replay: on its own wall-clock schedule it plays `args.eval_eps` whole episodes
  with `mode='eval'` and draws one report batch from each replay.
"""
import pathlib
from functools import partial as bind

from .. import utils
from ..core.driver import Driver
from .train import _Learner


class _Evaluation:
  """Episodes in eval mode into their own Replay, plus the two report streams."""

  def __init__(self, agent, make_env, replay_eval, replay_train, make_stream, logger, args):
    self.agent, self.logger = agent, logger
    self.replays = {'eval': replay_eval, 'report': replay_train}
    self.episodes = getattr(args, 'eval_eps', 1)
    count = getattr(args, 'eval_envs', 1)
    self.driver = Driver(
        [bind(make_env, index) for index in range(count)],
        parallel=not args.debug, device=getattr(args, 'device', None))
    self.driver.on_step(replay_eval.add)
    self.streams = {
        'eval': iter(agent.stream(make_stream(replay_eval, 'eval'))),
        'report': iter(agent.stream(make_stream(replay_train, 'report')))}
    self.carries = {name: agent.init_report(args.batch_size) for name in self.streams}
    self.due = utils.LocalClock(getattr(args, 'eval_every', args.report_every), first=True)

  def policy(self, *args, **kwargs):
    return self.agent.policy(*args, mode='eval', **kwargs)

  def maybe_run(self, step):
    if not self.due(step):
      return
    self.driver.reset(self.agent.init_policy)
    self.driver(self.policy, episodes=self.episodes)
    for name, stream in self.streams.items():
      if len(self.replays[name]):
        self.carries[name], metrics = self.agent.report(self.carries[name], next(stream))
        self.logger.add(metrics, prefix=name)

  def close(self):
    self.driver.close()


def train_eval(
    make_agent, make_replay_train, make_replay_eval, make_env_train,
    make_env_eval, make_stream, make_logger, args):
  agent, logger = make_agent(), make_logger()
  replays = {'train': make_replay_train(), 'eval': make_replay_eval()}
  step = logger.step
  policy_fps = utils.FPS()
  log_due = utils.LocalClock(args.log_every)
  save_due = utils.LocalClock(args.save_every)

  actors = Driver(
      [bind(make_env_train, index) for index in range(args.envs)],
      parallel=not args.debug, device=getattr(args, 'device', None))

  def count(trans, workers, **kw):
    step.increment(args.envs)
    policy_fps.step(args.envs)

  actors.on_batch(count)
  actors.on_step(replays['train'].add)
  learner = _Learner(
      agent, replays['train'], iter(agent.stream(make_stream(replays['train'], 'train'))),
      step, args)
  actors.on_batch(learner)
  evaluation = _Evaluation(
      agent, make_env_eval, replays['eval'], replays['train'], make_stream, logger, args)

  checkpoint = utils.Checkpoint(pathlib.Path(args.logdir) / 'checkpoint.pkl')
  checkpoint.step = step
  checkpoint.agent = agent
  checkpoint.replay_train = replays['train']
  checkpoint.replay_eval = replays['eval']
  checkpoint.load_or_save()

  def policy(*a, **kw):
    return agent.policy(*a, mode='train', **kw)

  actors.reset(agent.init_policy)
  while step < args.steps:
    evaluation.maybe_run(step)
    actors(policy, steps=10)
    if log_due(step):
      logger.add(learner.metrics.result())
      logger.add(replays['train'].stats(), prefix='replay')
      logger.add({'fps/policy': policy_fps.result(), 'fps/train': learner.fps.result()})
      logger.write()
    if save_due(step):
      checkpoint.save()
  logger.close()
  actors.close()
  evaluation.close()
