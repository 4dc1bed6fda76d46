"""Blocking predicate wait and the samples-per-insert rate limiter
(reference: embodied/core/limiters.py:5-80)."""
import threading
import time


def wait(predicate, message, info=None, sleep=0.01, notify=60):
  """Spin until `predicate()`; returns the seconds waited (limiters.py:5-16)."""
  if predicate():
    return 0
  began = reported = time.time()
  while not predicate():
    now = time.time()
    if now - reported > notify:
      print(f'{message} {now - began:.1f}s: {info}')
      reported = now
    time.sleep(sleep)
  return time.time() - began


class SamplesPerInsert:
  """Keeps sampling within `tolerance` of `samples_per_insert` x inserts once
  `minsize` inserts happened (limiters.py:19-80)."""

  def __init__(self, samples_per_insert, tolerance, minsize):
    assert 1 <= minsize
    self.samples_per_insert = samples_per_insert
    self.minsize = minsize
    self.avail = -minsize
    self.min_avail = -tolerance
    self.max_avail = tolerance * samples_per_insert
    self.size = 0
    self.lock = threading.Lock()

  def save(self):
    return {'size': self.size, 'avail': self.avail}

  def load(self, data):
    self.size = data['size']
    self.avail = data['avail']

  def want_insert(self):
    if self.size < self.minsize or self.samples_per_insert <= 0:
      return True
    return self.avail < self.max_avail

  def want_sample(self):
    if self.size < self.minsize:
      return False
    if self.samples_per_insert <= 0:
      return True
    return self.min_avail < self.avail

  def insert(self):
    with self.lock:
      self.size += 1
      if self.size >= self.minsize:
        self.avail += self.samples_per_insert

  def sample(self):
    with self.lock:
      self.avail -= 1
