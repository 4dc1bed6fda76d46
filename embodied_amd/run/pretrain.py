"""Learner-only loop: train steps from replay streams, no environment.

Same entry point and `args` fields as the reference's
`embodied.run.pretrain(make_model, make_stream, make_logger, args)`
(embodied/run/pretrain.py:8-96): `make_stream(None, mode)` builds the stream of
a replay that was filled elsewhere (loaded from its directory), every iteration
is `next(stream)` -> `model.train`; on the report schedule `report_batches *
consec_report` batches of the report stream and of the eval stream go through
`model.report`; log / report / save fire on GlobalClocks (in lockstep over the
ranks of a `torch.distributed` job, one process: wall clock).  This is the
learner half of the metric by itself: sample gather + return scan per step.
"""
import pathlib
import pickle
import time

from .. import utils
from .. import distributed


def _restorable(value):
  return hasattr(value, 'save') and hasattr(value, 'load')


def pretrain(make_model, make_stream, make_logger, args):
  model = make_model()
  streams = {
      mode: iter(model.stream(make_stream(None, mode))) for mode in ('train', 'report', 'eval')}
  logger = make_logger()
  step = logger.step
  should = {
      name: distributed.GlobalClock(getattr(args, f'{name}_every'))
      for name in ('log', 'report', 'save')}
  metrics = utils.Agg()
  fps = utils.FPS()
  batch_steps = args.batch_size * args.batch_length
  carry = model.init_train(args.batch_size)
  carries = {'report': model.init_report(args.batch_size), 'eval': model.init_report(args.batch_size)}

  checkpoint = utils.Checkpoint(pathlib.Path(args.logdir) / 'checkpoint.pkl')
  checkpoint.step = step
  checkpoint.model = model
  for mode, stream in streams.items():
    if _restorable(stream):                  # (the reference checkpoints its three iterators)
      setattr(checkpoint, f'dataset_{mode}', stream)
  if checkpoint.exists():
    checkpoint.load()
  else:
    if getattr(args, 'from_checkpoint', ''):
      data = pickle.loads(pathlib.Path(args.from_checkpoint).read_bytes())
      regex = getattr(args, 'from_checkpoint_regex', None)
      model.load(data['model'], **({'regex': regex} if regex else {}))
    if getattr(args, 'replica', 0) == 0:
      checkpoint.save()

  def report(mode):
    agg = utils.Agg()
    start = time.time()
    for _ in range(args.consec_report * args.report_batches):
      carries[mode], mets = model.report(carries[mode], next(streams[mode]))
      agg.add(mets)
    logger.add({f'dur/{mode}': time.time() - start})
    logger.add(agg.result(), prefix=mode)

  print('Starting training')
  while step < args.steps:
    with utils.timer.section('stream'):
      batch = next(streams['train'])
    with utils.timer.section('train'):
      start = time.time()
      carry, outs, mets = model.train(carry, batch)
      logger.add({'dur/train': time.time() - start})
    metrics.add(mets)
    step.increment()
    fps.step(batch_steps)

    if should['report'](step):
      logger.write()
      print('Train report')
      report('report')
      print('Eval report')
      report('eval')
      logger.add({'timer': utils.timer.stats()['summary']})
      logger.write()

    if should['log'](step):
      logger.add(metrics.result(), prefix='train')
      rate = fps.result()
      logger.add({'fps': rate, 'spf': 1 / rate if rate else float('inf')})

    if should['save'](step) and getattr(args, 'replica', 0) == 0:
      checkpoint.save()

  logger.close()
