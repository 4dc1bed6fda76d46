"""Single-process actor + learner loop — the primary caller of the hot path.

Same entry point and `args` fields as the reference's
`embodied.run.train(make_agent, make_replay, make_env, make_stream,
make_logger, args)` (embodied/run/train.py:10-118): a Driver steps `args.envs`
environments and feeds the Replay; the learner runs inside a Driver callback,
`Ratio(train_ratio / (batch_size * batch_length))` train steps per env step,
each `next(stream)` -> `agent.train` -> optional `replay.update`; reports,
logs and checkpoints fire on wall-clock schedules.
"""
import pathlib
import pickle
from functools import partial as bind

from .. import utils
from ..core.driver import Driver
from .stats import EpisodeStats


class _Learner:
  """The train-step callback registered on the Driver."""

  def __init__(self, agent, replay, stream, step, args):
    self.agent, self.replay, self.stream = agent, replay, stream
    self.step = step
    self.batch_steps = args.batch_size * args.batch_length
    self.min_items = args.batch_size * args.batch_length
    self.ratio = utils.Ratio(args.train_ratio / self.batch_steps)
    self.carry = agent.init_train(args.batch_size)
    self.metrics = utils.Agg()
    self.fps = utils.FPS()

  def __call__(self, trans, workers=None, **kwargs):
    # Per env in the reference (so the Ratio sees every env step); evaluated
    # once per vectorised step here with the same counter: same repeat total.
    if len(self.replay) < self.min_items:
      return
    for _ in range(self.ratio(self.step)):
      with utils.timer.section('stream_next'):
        batch = next(self.stream)
      self.carry, outs, mets = self.agent.train(self.carry, batch)
      self.fps.step(self.batch_steps)
      if 'replay' in outs:
        self.replay.update(outs['replay'])
      self.metrics.add(mets, prefix='train')


def train(make_agent, make_replay, make_env, make_stream, make_logger, args):
  agent, replay, logger = make_agent(), make_replay(), make_logger()
  step = logger.step
  logdir = pathlib.Path(args.logdir)
  epstats = utils.Agg()
  policy_fps = utils.FPS()
  clocks = {
      name: utils.LocalClock(getattr(args, f'{name}_every'))
      for name in ('log', 'report', 'save')}

  driver = Driver(
      [bind(make_env, index) for index in range(args.envs)],
      parallel=not args.debug, device=getattr(args, 'device', None))
  episodes = EpisodeStats(logger, epstats)
  n = args.envs

  def count(trans, workers, **kw):
    step.increment(n)
    policy_fps.step(n)

  driver.on_batch(count)
  driver.on_step(replay.add)            # batched when the Driver is on a GPU
  driver.on_batch(episodes.on_batch)

  learner = _Learner(
      agent, replay, iter(agent.stream(make_stream(replay, 'train'))), step, args)
  driver.on_batch(learner)
  stream_report = iter(agent.stream(make_stream(replay, 'report')))
  carry_report = agent.init_report(args.batch_size)

  checkpoint = utils.Checkpoint(logdir / 'checkpoint.pkl')
  checkpoint.step = step
  checkpoint.agent = agent
  checkpoint.replay = replay
  if getattr(args, 'from_checkpoint', ''):
    data = pickle.loads(pathlib.Path(args.from_checkpoint).read_bytes())
    agent.load(data['agent'])
  checkpoint.load_or_save()

  print('Start training loop')
  policy = lambda *a, **kw: agent.policy(*a, mode='train', **kw)
  driver.reset(agent.init_policy)
  while step < args.steps:
    driver(policy, steps=10)

    if clocks['report'](step) and len(replay):
      agg = utils.Agg()
      for _ in range(args.consec_report * args.report_batches):
        carry_report, mets = agent.report(carry_report, next(stream_report))
        agg.add(mets)
      logger.add(agg.result(), prefix='report')

    if clocks['log'](step):
      logger.add(learner.metrics.result())
      logger.add(epstats.result(), prefix='epstats')
      logger.add(replay.stats(), prefix='replay')
      logger.add({
          'fps/policy': policy_fps.result(), 'fps/train': learner.fps.result(),
          'timer': utils.timer.stats()['summary']})
      logger.write()

    if clocks['save'](step):
      checkpoint.save()

  episodes.flush()
  # (The reference stops here and drops what was aggregated since the last
  # wall-clock log; a run shorter than `log_every` would leave no train metrics.)
  rest = learner.metrics.result()
  if rest:
    logger.add(rest)
    logger.write()
  logger.close()
  driver.close()
