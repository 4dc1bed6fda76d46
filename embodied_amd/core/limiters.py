"""Blocking wait and the samples-per-insert rate limiter
(reference: embodied/core/limiters.py:5-80)."""
import threading
import time


def wait(predicate, message, info=None, sleep=0.01, notify=60):
  """Poll `predicate` every `sleep` seconds until it holds; print `message`
  every `notify` seconds meanwhile.  Returns the seconds spent waiting (0 if it
  held at once).  Replay.sample uses it to block on an empty buffer."""
  began = time.time()
  reminded = began
  waited = False
  while not predicate():
    waited = True
    now = time.time()
    if now - reminded > notify:
      print(f'{message} {now - began:.1f}s: {info}')
      reminded = now
    time.sleep(sleep)
  return time.time() - began if waited else 0


class SamplesPerInsert:
  """Token bucket coupling the learner's sampling rate to the actors' insert
  rate: each insert (once `minsize` items exist) adds `samples_per_insert`
  tokens, each sample takes one; inserting pauses when the surplus exceeds
  `tolerance * samples_per_insert`, sampling when the deficit exceeds
  `tolerance`.  State = (size, avail), checkpointable."""

  def __init__(self, samples_per_insert, tolerance, minsize):
    if minsize < 1:
      raise AssertionError(minsize)
    self.samples_per_insert = samples_per_insert
    self.minsize = minsize
    self.size = 0
    self.avail = -minsize
    self.min_avail = -tolerance
    self.max_avail = tolerance * samples_per_insert
    self.lock = threading.Lock()

  @property
  def _warm(self):
    return self.size >= self.minsize

  @property
  def _limited(self):
    return self.samples_per_insert > 0

  def want_insert(self):
    return (not self._warm) or (not self._limited) or self.avail < self.max_avail

  def want_sample(self):
    if not self._warm:
      return False
    return (not self._limited) or self.avail > self.min_avail

  def insert(self):
    with self.lock:
      self.size += 1
      if self._warm:
        self.avail += self.samples_per_insert

  def sample(self):
    with self.lock:
      self.avail -= 1

  def save(self):
    return dict(size=self.size, avail=self.avail)

  def load(self, data):
    self.size, self.avail = data['size'], data['avail']
