"""Run-loop utilities (stand-ins for the used surface of `elements`)."""
import time

import numpy as np

from embodied_amd import utils


def test_ratio_matches_reference_schedule():
  """when.Ratio [SURVEY App. A]: first call 1, then int((step-prev)*ratio) with
  the remainder carried; 64-env steps at train_ratio 3 / (16*64)."""
  ratio = utils.Ratio(3.0 / 1024)
  calls = [ratio(s) for s in range(0, 64 * 2000, 64)]
  assert calls[0] == 1 and set(calls[1:]) <= {0, 1}
  assert abs(sum(calls) - (64 * 1999 * 3 / 1024 + 1)) <= 1
  assert utils.Ratio(0)(5) == 0
  fast = utils.Ratio(2.5)
  assert [fast(s) for s in (0, 1, 2, 3, 4)] == [1, 2, 3, 2, 3]


def test_counter_behaves_like_an_int():
  c = utils.Counter()
  c.increment(); c.increment(4)
  assert int(c) == 5 and c >= 5 and c < 6 and c - 2 == 3 and c == 5
  c.load(c.save() + 1)
  assert int(c) == 6


def test_agg_modes():
  agg = utils.Agg()
  for v in (1.0, 3.0):
    agg.add('a', v, agg='sum')
    agg.add('b', v, agg=('avg', 'max'))
    agg.add({'c': v}, prefix='p')
    agg.add('d', np.array([v, v]), agg='stack')
  out = agg.result()
  assert out['a'] == 4 and out['b/avg'] == 2 and out['b/max'] == 3 and out['p/c'] == 2
  assert out['d'].shape == (2, 2)
  assert agg.result() == {}


def test_local_clock():
  clock = utils.LocalClock(0.05)
  assert clock() is False            # first call arms it
  time.sleep(0.06)
  assert clock() is True and clock() is False
  assert utils.LocalClock(0)() is False and utils.LocalClock(-1)() is True


def test_checkpoint_roundtrip(tmp_path):
  class Thing:
    def __init__(self):
      self.state = 0
    def save(self):
      return {'state': self.state}
    def load(self, data):
      self.state = data['state']
  a, step = Thing(), utils.Counter(7)
  a.state = 42
  cp = utils.Checkpoint(tmp_path / 'cp.pkl')
  cp.thing, cp.step = a, step
  assert not cp.exists()
  cp.load_or_save()
  assert cp.exists()
  b, step2 = Thing(), utils.Counter()
  cp2 = utils.Checkpoint(tmp_path / 'cp.pkl')
  cp2.thing, cp2.step = b, step2
  cp2.load_or_save()
  assert b.state == 42 and int(step2) == 7


def test_logger_and_timer():
  lines = []
  logger = utils.Logger(printer=lines.append)
  logger.step.increment(3)
  logger.add({'x': 1.5}, prefix='m')
  logger.write()
  assert logger.history == [{'step': 3, 'm/x': 1.5}] and 'm/x' in lines[0]
  utils.timer.enabled = True
  with utils.timer.section('work'):
    pass
  stats = utils.timer.stats()
  utils.timer.enabled = False
  assert stats['work']['count'] == 1 and 'work' in stats['summary']
