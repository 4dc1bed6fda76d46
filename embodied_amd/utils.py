"""Small run-loop utilities.  The reference takes these from the un-vendored
`elements` package (Counter, Agg, FPS, when.Ratio, when.Clock, Checkpoint,
timer; used at embodied/run/train.py:16-29,82-89).  Only the behaviour the run
loops rely on is provided [SURVEY.md Appendix A]."""
import collections
import contextlib
import pathlib
import pickle
import threading
import time

import numpy as np


class Counter:

  def __init__(self, initial=0):
    self.value = initial
    self.lock = threading.Lock()

  def __int__(self):
    return int(self.value)

  __index__ = __int__

  def __eq__(self, other):
    return int(self) == other

  def __lt__(self, other):
    return int(self) < other

  def __le__(self, other):
    return int(self) <= other

  def __gt__(self, other):
    return int(self) > other

  def __ge__(self, other):
    return int(self) >= other

  def __sub__(self, other):
    return int(self) - other

  def __repr__(self):
    return f'Counter({self.value})'

  def increment(self, amount=1):
    with self.lock:
      self.value += amount

  def reset(self):
    with self.lock:
      self.value = 0

  def save(self):
    return self.value

  def load(self, value):
    self.value = value


class Ratio:
  """How many times to repeat per elapsed steps: first call 1, afterwards
  int((step - prev) * ratio) with the remainder carried (when.Ratio)."""

  def __init__(self, ratio):
    assert ratio >= 0, ratio
    self.ratio = ratio
    self.prev = None

  def __call__(self, step):
    step = int(step)
    if self.ratio == 0:
      return 0
    if self.prev is None:
      self.prev = step
      return 1
    repeats = int((step - self.prev) * self.ratio)
    self.prev += repeats / self.ratio
    return repeats


class LocalClock:
  """True at most once per `every` seconds (clock.py:97-118)."""

  def __init__(self, every, first=False):
    self.every = every
    self.prev = None
    self.first = first

  def __call__(self, step=None, skip=None):
    if skip:
      return False
    if self.every == 0:
      return False
    if self.every < 0:
      return True
    now = time.time()
    if self.prev is None:
      self.prev = now
      return self.first
    if now >= self.prev + self.every:
      self.prev = now
      return True
    return False


class FPS:

  def __init__(self):
    self.start = time.time()
    self.total = 0

  def step(self, amount=1):
    self.total += amount

  def result(self, reset=True):
    now = time.time()
    fps = self.total / max(now - self.start, 1e-9)
    if reset:
      self.start, self.total = now, 0
    return fps


class Agg:
  """Keyed aggregation: sum / avg / max / stack (+ tuples of them)."""

  def __init__(self):
    self.reset()

  def reset(self):
    self.store = {}
    self.how = {}

  def add(self, key_or_map, value=None, agg='default', prefix=None):
    if value is not None:
      mapping = {key_or_map: value}
    else:
      mapping = key_or_map
    for key, value in mapping.items():
      key = f'{prefix}/{key}' if prefix else key
      how = agg
      if how == 'default':
        how = 'avg' if np.ndim(value) == 0 else 'last'
      self.how[key] = how
      self.store.setdefault(key, []).append(np.asarray(value))

  def result(self, reset=True, prefix=None):
    out = {}
    for key, values in self.store.items():
      hows = self.how[key]
      multi = isinstance(hows, (tuple, list))
      for how in (hows if multi else (hows,)):
        name = f'{key}/{how}' if multi else key
        if how == 'sum':
          out[name] = np.sum(values, 0)
        elif how == 'avg':
          out[name] = np.mean(values, 0)
        elif how == 'max':
          out[name] = np.max(values, 0)
        elif how == 'stack':
          out[name] = np.stack(values)
        else:
          out[name] = values[-1]
    if prefix:
      out = {f'{prefix}/{k}': v for k, v in out.items()}
    reset and self.reset()
    return out


class Timer:
  """Named wall-clock sections (decorator or context), the instrument the
  reference wraps around every hot function (elements.timer.section)."""

  def __init__(self):
    self.enabled = False
    self.stats_ = collections.defaultdict(lambda: [0, 0.0])

  @contextlib.contextmanager
  def section(self, name):
    if not self.enabled:
      yield
      return
    start = time.perf_counter()
    try:
      yield
    finally:
      entry = self.stats_[name]
      entry[0] += 1
      entry[1] += time.perf_counter() - start

  def stats(self, reset=True):
    out = {k: {'count': c, 'sum': s, 'avg': s / max(c, 1)} for k, (c, s) in self.stats_.items()}
    summary = ' | '.join(f'{k}: {v["sum"]:.2f}s/{v["count"]}' for k, v in sorted(out.items()))
    if reset:
      self.stats_.clear()
    return {'summary': summary, **out}


timer = Timer()


class Checkpoint:
  """Attributes assigned to the checkpoint are saved/loaded through their own
  save()/load() (run/train.py:82-89)."""

  def __init__(self, filename=None):
    object.__setattr__(self, '_filename', filename and pathlib.Path(filename))
    object.__setattr__(self, '_values', {})

  def __setattr__(self, name, value):
    if name.startswith('_'):
      return object.__setattr__(self, name, value)
    assert hasattr(value, 'save') and hasattr(value, 'load'), name
    self._values[name] = value

  def __getattr__(self, name):
    try:
      return object.__getattribute__(self, '_values')[name]
    except KeyError:
      raise AttributeError(name)

  def exists(self):
    return bool(self._filename and self._filename.exists())

  def save(self, filename=None):
    filename = pathlib.Path(filename) if filename else self._filename
    data = {k: v.save() for k, v in self._values.items()}
    filename.parent.mkdir(parents=True, exist_ok=True)
    tmp = filename.with_suffix('.tmp')
    tmp.write_bytes(pickle.dumps(data))
    tmp.replace(filename)

  def load(self, filename=None, keys=None):
    filename = pathlib.Path(filename) if filename else self._filename
    data = pickle.loads(filename.read_bytes())
    for key in keys or data.keys():
      self._values[key].load(data[key])

  def load_or_save(self):
    if self.exists():
      self.load()
    else:
      self.save()


class Logger:
  """Minimal metric sink with the surface the run loops use: `.step`,
  `.add(mapping, prefix=)`, `.write()`, `.close()`."""

  def __init__(self, step=None, printer=None):
    self.step = step if step is not None else Counter()
    self.printer = printer
    self.pending = {}
    self.history = []

  def add(self, mapping, prefix=None):
    for key, value in mapping.items():
      key = f'{prefix}/{key}' if prefix else key
      self.pending[key] = value

  def write(self):
    if not self.pending:
      return
    record = {'step': int(self.step), **self.pending}
    self.history.append(record)
    if self.printer:
      scalars = {k: float(v) for k, v in record.items() if np.ndim(v) == 0 and not isinstance(v, str)}
      self.printer(' / '.join(f'{k} {v:.4g}' for k, v in scalars.items()))
    self.pending = {}

  def close(self):
    self.write()
