"""Per-episode statistics as a vectorised reducer.

The reference computes score / length / reward_rate (and 'log/*' aggregates)
in a per-env Python callback that runs N times per step
(embodied/run/train.py:31-54).  Here one `on_batch` call per vectorised step
updates (N,) accumulators; device transitions are read back through small
pinned buffers one step late, so the Driver never waits for the GPU.
"""
import collections

import numpy as np
import torch


class EpisodeStats:

  def __init__(self, logger, epstats, lag=2):
    self.logger = logger
    self.epstats = epstats
    self.lag = lag
    self.inflight = collections.deque()
    self.score = None

  def _alloc(self, n, log_keys):
    self.score = np.zeros(n, np.float64)
    self.length = np.zeros(n, np.int64)
    self.changes = np.zeros(n, np.int64)     # |r_t - r_{t-1}| >= 0.01 count
    self.prev_reward = np.zeros(n, np.float64)
    self.logs = {k: dict(sum=np.zeros(n), max=np.full(n, -np.inf)) for k in log_keys}

  def on_batch(self, trans, workers, **kwargs):
    keys = ['reward', 'is_first', 'is_last'] + sorted(
        k for k in trans if k.startswith('log/'))
    values = [trans[k] for k in keys]
    if any(torch.is_tensor(v) and v.is_cuda for v in values):
      host = [torch.empty(v.shape, dtype=v.dtype).pin_memory() for v in values]
      for dst, src in zip(host, values):
        dst.copy_(src, non_blocking=True)
      event = torch.cuda.Event()
      event.record()
      self.inflight.append((keys, host, event))
      while self.inflight and (
          len(self.inflight) > self.lag or self.inflight[0][2].query()):
        keys_, host_, event_ = self.inflight.popleft()
        event_.synchronize()
        self._reduce(dict(zip(keys_, [h.numpy() for h in host_])))
    else:
      self._reduce({k: np.asarray(v.cpu() if torch.is_tensor(v) else v)
                    for k, v in zip(keys, values)})

  def flush(self):
    while self.inflight:
      keys, host, event = self.inflight.popleft()
      event.synchronize()
      self._reduce(dict(zip(keys, [h.numpy() for h in host])))

  def _reduce(self, tran):
    reward = tran['reward'].astype(np.float64)
    first = tran['is_first'].astype(bool)
    last = tran['is_last'].astype(bool)
    log_keys = [k for k in tran if k.startswith('log/')]
    if self.score is None:
      self._alloc(len(reward), log_keys)
    # An episode restarts where is_first is set.
    self.score[first] = 0
    self.length[first] = 0
    self.changes[first] = 0
    for key in log_keys:
      self.logs[key]['sum'][first] = 0
      self.logs[key]['max'][first] = -np.inf
    moved = (np.abs(reward - self.prev_reward) >= 0.01) & (self.length > 0)
    self.changes += moved
    self.score += reward
    self.length += 1
    self.prev_reward = reward
    for key in log_keys:
      value = tran[key].astype(np.float64)
      self.logs[key]['sum'] += value
      self.logs[key]['max'] = np.maximum(self.logs[key]['max'], value)
    for env in np.flatnonzero(last):
      n = int(self.length[env])
      self.logger.add(
          {'score': self.score[env], 'length': n}, prefix='episode')
      result = {}
      if n > 1:
        result['reward_rate'] = self.changes[env] / (n - 1)
      for key in log_keys:
        total = self.logs[key]['sum'][env]
        result[f'{key}/avg'] = total / n
        result[f'{key}/max'] = self.logs[key]['max'][env]
        result[f'{key}/sum'] = total
      self.epstats.add(result)
