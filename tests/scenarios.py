"""Scripted scenarios shared by the golden generator, the oracle tests and the
product (HIP) parity tests.

Each scenario is a function of a namespace `ns` exposing the reference-shaped
API (`Replay`, `Uniform`, `Prioritized`, `Mixture`, `SampleTree`, `Consec`,
`Driver`, `tonp`) and returns `{name: ndarray}`.  `oracle/gen_golden.py` runs
them against the real reference (build container only) and commits the result
under `tests/golden/`; tests run them against `oracle/np_oracle.py` and against
`embodied_amd` and compare with those files.
"""
import numpy as np


class Spec:
  """Minimal space: what Driver reads from `act_space` (driver.py:35-37)."""

  def __init__(self, dtype, shape=()):
    self.dtype = np.dtype(dtype)
    self.shape = tuple(shape)


class ScriptEnv:
  """Deterministic fixture env modelled on the reference's Dummy
  (embodied/envs/dummy.py:38-59): episode of `length` steps, reset on request
  or after the terminal step; observations are functions of (env id, count)."""

  def __init__(self, ident, length, image=(4, 4, 2)):
    self.ident = ident
    self.length = length
    self.image = image
    self.count = 0
    self.done = False
    self.total = 0

  @property
  def act_space(self):
    return {
        'reset': Spec(bool),
        'act_disc': Spec(np.int32),
        'act_cont': Spec(np.float32, (3,)),
    }

  @property
  def obs_space(self):
    return {
        'image': Spec(np.uint8, self.image),
        'count': Spec(np.float32),
        'reward': Spec(np.float32),
        'is_first': Spec(bool),
        'is_last': Spec(bool),
        'is_terminal': Spec(bool),
    }

  def step(self, action):
    action = dict(action)
    self.total += 1
    if action.pop('reset') or self.done:
      self.count = 0
      self.done = False
      return self._obs(0.0, is_first=True)
    self.count += 1
    self.done = self.count >= self.length
    rew = float(action['act_disc']) + 0.5 * self.ident
    return self._obs(rew, is_last=self.done, is_terminal=self.done and self.ident % 2 == 0)

  def _obs(self, reward, is_first=False, is_last=False, is_terminal=False):
    n = int(np.prod(self.image))
    image = ((np.arange(n) * 3 + self.count * 7 + self.ident * 31) % 256)
    return {
        'image': image.astype(np.uint8).reshape(self.image),
        'count': np.float32(self.count),
        'reward': np.float32(reward),
        'is_first': is_first,
        'is_last': is_last,
        'is_terminal': is_terminal,
        'log/total': np.float32(self.total),
    }

  def close(self):
    pass


def synth_step(t, w, image=(6, 5, 3)):
  """A replay step whose every byte is a function of (t, w)."""
  n = int(np.prod(image))
  return {
      'image': ((np.arange(n) + 13 * t + 101 * w) % 251).astype(np.uint8).reshape(image),
      'step': np.int32(t),
      'worker': np.int32(w),
      'reward': np.float32(0.25 * t - w),
      'vec': (np.arange(5) * 0.5 + t + 100 * w).astype(np.float32),
      'is_first': np.bool_(t % 11 == 0),
      'is_last': np.bool_(t % 11 == 10),
      'is_terminal': np.bool_(t % 22 == 10),
  }


# ---------------------------------------------------------------- selectors --


def sel_uniform(ns):
  out = {}
  u = ns.Uniform(0)
  for k in range(10):
    u[k] = None
  out['draws_a'] = np.array([u() for _ in range(16)])
  del u[3]
  out['draws_b'] = np.array([u() for _ in range(8)])
  # capacity churn: FIFO-evict the oldest while inserting, as Replay does.
  u = ns.Uniform(7)
  live, draws = [], []
  for k in range(200):
    if len(live) >= 37:
      del u[live.pop(0)]
    u[k] = None
    live.append(k)
    if k % 3 == 0:
      draws.append(u())
  out['draws_churn'] = np.array(draws)
  # n == 1 consumes no randomness, then mixed sizes (numpy Lemire path).
  u = ns.Uniform(123)
  u[100] = None
  draws = [u() for _ in range(3)]
  for k in range(1, 300):
    u[100 + k] = None
    draws.append(u())
  out['draws_grow'] = np.array(draws)
  return out


def sel_sampletree(ns):
  out = {}
  for branching in (2, 3, 16):
    tree = ns.SampleTree(branching, seed=0)
    for k in range(40):
      tree.insert(k, float(k % 5))
    a = [tree.sample() for _ in range(16)]
    for k in range(0, 40, 3):
      tree.remove(k)
    b = [tree.sample() for _ in range(16)]
    for k in range(40, 60):
      tree.insert(k, 0.1 * (k % 7) + 1e-3)
    for k in range(1, 40, 6):
      tree.update(k, 2.5 + k / 7)
    c = [tree.sample() for _ in range(32)]
    out[f'b{branching}'] = np.array(a + b + c)
  tree = ns.SampleTree(16, seed=5)
  for k in range(100):
    tree.insert(k, np.inf if k % 3 == 0 else 1.0)
  out['inf'] = np.array([tree.sample() for _ in range(32)])
  tree = ns.SampleTree(4, seed=6)
  for k in range(30):
    tree.insert(k, 0.0)
  out['zero'] = np.array([tree.sample() for _ in range(32)])
  rng = np.random.default_rng(11)
  tree = ns.SampleTree(5, seed=9)
  live, draws = [], []
  for k in range(400):
    op = rng.integers(0, 4)
    if op <= 1 or len(live) < 3:
      tree.insert(k, float(rng.random() * 3))
      live.append(k)
    elif op == 2:
      victim = live.pop(int(rng.integers(0, len(live))))
      tree.remove(victim)
    else:
      tree.update(live[int(rng.integers(0, len(live)))], float(rng.random()))
    draws.append(tree.sample())
  out['random_ops'] = np.array(draws)
  return out


def _sid(k):
  return np.frombuffer(int(k).to_bytes(20, 'big'), np.uint8)


def sel_prioritized(ns):
  out = {}
  for name, kw in {
      'plain': dict(exponent=1.0, initial=1.0, maxfrac=0.0, seed=0),
      'expmax': dict(exponent=0.8, initial=1.0, maxfrac=0.5, seed=0),
      'infzero': dict(exponent=0.8, initial=np.inf, maxfrac=0.5,
                      zero_on_sample=True, seed=3),
  }.items():
    sel = ns.Prioritized(**kw)
    length, draws = 4, []
    for k in range(24):           # item k covers steps k..k+3 (overlapping)
      sel[k] = np.stack([_sid(s) for s in range(k, k + length)])
    draws += [sel() for _ in range(12)]
    steps = np.stack([_sid(s) for s in range(3, 19)])
    sel.prioritize(steps, np.linspace(0.1, 4.0, len(steps)))
    draws += [sel() for _ in range(12)]
    for k in range(0, 8):
      del sel[k]
    for k in range(24, 30):
      sel[k] = np.stack([_sid(s) for s in range(k, k + length)])
    sel.prioritize(np.stack([_sid(s) for s in (25, 26, 9)]), np.array([7.0, 0.0, 2.0]))
    draws += [sel() for _ in range(24)]
    # Priorities for step ids that belong to NO item at that moment (selectors.py:143-150
    # takes them: `prios` is a defaultdict): 33..36 arrive with later items and keep what
    # they were given; 0 lost its last item above; 31, 32 are in items 28, 29 already.
    early = (31, 32, 33, 34, 35, 36, 0)
    sel.prioritize(np.stack([_sid(s) for s in early]), np.array([0.5, 3.0, 6.0, 0.0, 2.5, 9.0, 4.0]))
    draws += [sel() for _ in range(8)]
    for k in range(30, 34):         # items 30..33 cover steps 30..36
      sel[k] = np.stack([_sid(s) for s in range(k, k + length)])
    draws += [sel() for _ in range(24)]
    sel.prioritize(np.stack([_sid(s) for s in (35, 37)]), np.array([0.25, 5.0]))   # 37: early again
    del sel[8]
    sel[34] = np.stack([_sid(s) for s in range(34, 34 + length)])
    draws += [sel() for _ in range(16)]
    out[name] = np.array(draws)
  return out


def sel_mixture(ns):
  uni = ns.Uniform(1)
  pri = ns.Prioritized(exponent=0.8, initial=1.0, maxfrac=0.5, seed=2)
  mix = ns.Mixture(
      dict(uniform=uni, priority=pri), dict(uniform=0.5, priority=0.5), seed=3)
  for k in range(8):
    mix[k] = np.stack([_sid(s) for s in range(k, k + 3)])
  a = [mix() for _ in range(16)]
  mix.prioritize(np.stack([_sid(s) for s in range(2, 6)]), np.array([5.0, 0.5, 3.0, 0.0]))
  del mix[0]
  b = [mix() for _ in range(16)]
  return {'draws': np.array(a + b)}


# ------------------------------------------------------------------- replay --


def _dump(ns, batch, prefix, out):
  for k, v in batch.items():
    out[f'{prefix}/{k}'] = ns.tonp(v)


def replay_basic(ns):
  """SURVEY Appendix C capture: length 5, capacity 50, chunksize 8, 3 workers."""
  out = {}
  rep = ns.Replay(length=5, capacity=50, chunksize=8, seed=0)
  lens = []
  for t in range(30):
    for w in range(3):
      rep.add(synth_step(t, w), worker=w)
    lens.append(len(rep))
  out['lens'] = np.array(lens)
  _dump(ns, rep.sample(4), 's0', out)
  _dump(ns, rep.sample(7, 'report'), 's1', out)
  stats = rep.stats()
  out['stats'] = np.array(
      [stats[k] for k in ('items', 'chunks', 'streams', 'inserts', 'samples', 'updates')],
      np.float64)
  return out


def replay_chunk_spans(ns):
  """chunksize smaller than length: windows span several chunks
  (tests/test_replay.py:58-73 uses chunksize=4, length=7)."""
  out = {}
  rep = ns.Replay(length=7, capacity=27, chunksize=4, seed=1)
  for t in range(40):
    for w in range(2):
      rep.add(synth_step(t, w), worker=w)
    if t in (9, 20, 39):
      _dump(ns, rep.sample(5), f't{t}', out)
  out['len'] = np.array(len(rep))
  return out


def replay_uneven_workers(ns):
  """Workers advance at different rates (tests/test_replay.py:119-135)."""
  out = {}
  rng = np.random.default_rng(0)
  rep = ns.Replay(length=4, capacity=30, chunksize=6, seed=2)
  clock = [0, 0, 0, 0]
  lens = []
  for n in range(160):
    w = int(rng.integers(0, 4))
    rep.add(synth_step(clock[w], w), worker=w)
    clock[w] += 1
    lens.append(len(rep))
    if n in (40, 100, 159):
      _dump(ns, rep.sample(6), f'n{n}', out)
  out['lens'] = np.array(lens)
  return out


def replay_online(ns):
  """Online queue (replay.py:114-118,158-160): train drains fresh windows in
  order, report never touches the queue."""
  out = {}
  rep = ns.Replay(length=4, capacity=20, chunksize=5, online=True, seed=0)
  for t in range(14):
    rep.add(synth_step(t, 0), worker=0)
  _dump(ns, rep.sample(2, 'report'), 'report', out)
  _dump(ns, rep.sample(3, 'train'), 'train_a', out)
  _dump(ns, rep.sample(2, 'train'), 'train_b', out)
  for t in range(14, 60):
    for w in range(2):
      rep.add(synth_step(t, w), worker=w)
  # capacity 20 evicted most queued windows: stale entries are skipped.
  _dump(ns, rep.sample(6, 'train'), 'train_c', out)
  return out


def replay_update(ns):
  """Write-back across chunk boundaries and onto evicted rows
  (replay.py:129-149, 216-235)."""
  out = {}
  rep = ns.Replay(length=6, capacity=12, chunksize=4, seed=4)
  for t in range(20):
    rep.add(synth_step(t, 0), worker=0)
  batch = rep.sample(5)
  stale = {k: ns.tonp(v).copy() for k, v in batch.items()}
  sid = ns.tonp(batch['stepid'])
  upd = {
      'stepid': ns.like(batch['stepid'], sid[:, :4]),
      'vec': ns.like(batch['vec'], -np.arange(5 * 4 * 5, dtype=np.float32).reshape(5, 4, 5)),
      'reward': ns.like(batch['reward'], 100 + np.arange(20, dtype=np.float32).reshape(5, 4)),
  }
  rep.update(upd)
  for t in range(20, 26):
    rep.add(synth_step(t, 0), worker=0)
  _dump(ns, rep.sample(8), 'after', out)
  # update aimed at rows whose first chunk is long gone: silently skipped.
  for t in range(26, 60):
    rep.add(synth_step(t, 0), worker=0)
  rep.update({
      'stepid': ns.like(batch['stepid'], stale['stepid']),
      'reward': ns.like(batch['reward'], np.full((5, 6), -7, np.float32)),
  })
  _dump(ns, rep.sample(8), 'late', out)
  return out


def replay_prioritized(ns):
  """Replay driving a Prioritized selector with priority feedback.  (The
  reference's Mixture has no __len__, so Replay.sample raises with it —
  replay.py:123 vs selectors.py:200-228 — Mixture is pinned standalone above.)"""
  out = {}
  sel = ns.Prioritized(
      exponent=0.8, initial=np.inf, maxfrac=0.5, zero_on_sample=True, seed=6)
  rep = ns.Replay(length=3, capacity=16, chunksize=5, selector=sel, seed=0)
  for t in range(12):
    for w in range(2):
      rep.add(synth_step(t, w), worker=w)
  for r in range(6):
    batch = rep.sample(4)
    out[f'r{r}/step'] = ns.tonp(batch['step'])
    out[f'r{r}/worker'] = ns.tonp(batch['worker'])
    prio = (ns.tonp(batch['step']) % 5 + 0.5 * r).astype(np.float32)
    rep.update({'stepid': batch['stepid'], 'priority': ns.like(batch['reward'], prio)})
    for w in range(2):
      rep.add(synth_step(12 + r, w), worker=w)
  return out


def replay_baseline_shape(ns):
  """BASELINE chunking and fan-in with a tiny payload: 64 workers in lockstep,
  length 65, chunksize 1024, online queue, capacity churn; windows cross the
  1024-row chunk boundary.  Pins the index streams at the benchmark's shape."""
  out = {}
  rep = ns.Replay(length=65, capacity=5000, chunksize=1024, online=True, seed=0)
  checkpoints = {400: 'a', 1100: 'b'}
  for t in range(1101):
    for w in range(64):
      rep.add({'t': np.int32(t), 'w': np.int32(w),
               'is_first': np.bool_((t + 13 * w) % 250 == 0),
               'is_last': np.bool_((t + 13 * w) % 250 == 249)}, worker=w)
    if t in checkpoints:
      tag = checkpoints[t]
      out[f'{tag}/len'] = np.array(len(rep))
      for i in range(3):
        batch = rep.sample(16, 'train')
        for k in ('t', 'w', 'is_first', 'is_last'):
          out[f'{tag}/train{i}/{k}'] = ns.tonp(batch[k])
        out[f'{tag}/train{i}/row'] = ns.tonp(batch['stepid'])[..., 16:]
      batch = rep.sample(16, 'report')
      out[f'{tag}/report/t'] = ns.tonp(batch['t'])
      out[f'{tag}/report/w'] = ns.tonp(batch['w'])
  return out


def stream_consec(ns):
  out = {}
  rep = ns.Replay(length=7, capacity=40, chunksize=8, seed=3)
  for t in range(30):
    rep.add(synth_step(t, 0), worker=0)
  stream = ns.consec(rep, batch=2, length=3, consec=2, prefix=1)
  for n in range(5):
    _dump(ns, next(stream), f'w{n}', out)
  return out


# ------------------------------------------------------------------- driver --


def driver_script(ns):
  out = {}
  envs = [ScriptEnv(i, length=3 + i) for i in range(3)]
  driver = ns.Driver(envs)
  seen, carries, log = [], [], []

  def policy(carry, obs, **kw):
    seen.append({k: ns.tonp(v).copy() for k, v in obs.items()})
    carries.append(carry)
    n = len(ns.tonp(obs['is_first']))
    tick = (carry or 0)
    act = {
        'act_disc': (np.arange(n) + tick + 1).astype(np.int32),
        'act_cont': (np.arange(n * 3).reshape(n, 3) * 0.5 + tick + 1).astype(np.float32),
    }
    outs = {'logp': (np.arange(n) * -0.1 - tick).astype(np.float32)}
    return tick + 1, act, outs

  driver.on_step(lambda tran, worker, **kw: log.append(
      (worker, {k: np.array(ns.tonp(v)) for k, v in tran.items()})))
  driver.reset(lambda n: 0)
  driver(policy, steps=30)
  out['n_calls'] = np.array(len(seen))
  out['workers'] = np.array([w for w, _ in log])
  for key in sorted(log[0][1]):
    out[f'tran/{key}'] = np.stack([t[key] for _, t in log])
  for key in sorted(seen[0]):
    out[f'obs/{key}'] = np.stack([o[key] for o in seen])
  out['carries'] = np.array(carries)
  log.clear()
  driver(policy, episodes=4)
  out['episodes_len'] = np.array(len(log))
  out['episodes_last'] = np.stack([t['is_last'] for _, t in log])
  return out


SCENARIOS = {
    'sel_uniform': sel_uniform,
    'sel_sampletree': sel_sampletree,
    'sel_prioritized': sel_prioritized,
    'sel_mixture': sel_mixture,
    'replay_basic': replay_basic,
    'replay_chunk_spans': replay_chunk_spans,
    'replay_uneven_workers': replay_uneven_workers,
    'replay_online': replay_online,
    'replay_update': replay_update,
    'replay_prioritized': replay_prioritized,
    'replay_baseline_shape': replay_baseline_shape,
    'stream_consec': stream_consec,
    'driver_script': driver_script,
}


# ----------------------------------------------------------------- wrappers --
# Host-only scenarios (no HBM path, no oracle restatement): the product is
# compared with the golden vectors recorded from the reference directly.


class _ScriptedEnv:
  """Continuous-control env whose spaces use every dtype family the boundary
  unifies, and that records the actions it really received."""

  def __init__(self, Space, episode=7):
    self._Space = Space
    self._episode = episode
    self._t = 0
    self.received = []

  @property
  def obs_space(self):
    S = self._Space
    return {
        'image': S(np.uint8, (2, 2, 1)),
        'vec': S(np.float64, (3,)),
        'count': S(np.int64, ()),
        'reward': S(np.float32, ()),
        'is_first': S(bool, ()),
        'is_last': S(bool, ()),
        'is_terminal': S(bool, ()),
    }

  @property
  def act_space(self):
    S = self._Space
    return {
        'action': S(np.float64, (2,), np.array([-2.0, 0.0]), np.array([2.0, 10.0])),
        'free': S(np.float32, (2,)),
        'choice': S(np.int64, (), 0, 4),
        'reset': S(bool, ()),
    }

  def step(self, action):
    if action['reset']:
      self._t = 0
    else:
      self._t += 1
    self.received.append({k: np.array(v) for k, v in action.items()})
    last = self._t >= self._episode
    return {
        'image': np.full((2, 2, 1), self._t % 256, np.uint8),
        'vec': np.asarray(action['action']).sum() + np.arange(3, dtype=np.float64) * self._t,
        'count': np.int64(self._t * 1000),
        'reward': np.float32(0.5 * self._t),
        'is_first': self._t == 0,
        'is_last': last,
        'is_terminal': last and self._t % 2 == 1,
    }


def _space_record(spaces):
  out = {}
  for key, space in spaces.items():
    out[f'{key}/dtype'] = np.frombuffer(np.dtype(space.dtype).str.encode().ljust(4), np.uint8).copy()
    out[f'{key}/shape'] = np.array(space.shape, np.int64)
    low, high = np.asarray(space.low), np.asarray(space.high)
    if np.issubdtype(space.dtype, np.integer):
      # How an out-of-range bound is stored is the Space type's business (the
      # reference takes Space from the un-vendored `elements`): compare the
      # bounds as far as the dtype can represent them.
      info = np.iinfo(space.dtype)
      low, high = np.clip(low, info.min, info.max), np.clip(high, info.min, info.max)
    out[f'{key}/low'] = np.asarray(low, np.float64)
    out[f'{key}/high'] = np.asarray(high, np.float64)
    out[f'{key}/discrete'] = np.array(bool(space.discrete))
  return out


def wrappers_chain(ns):
  """The reference's own `wrap_env` chain (ppo/main.py:249-258) plus TimeLimit
  and ActionRepeat: spaces seen outside, actions seen inside, observations, and
  which exception each malformed action raises."""
  W = ns.wrappers
  out = {}
  base = _ScriptedEnv(ns.Space)
  env = base
  for name, space in base.act_space.items():
    if not space.discrete:
      env = W.NormalizeAction(env, name)
  env = W.UnifyDtypes(env)
  env = W.CheckSpaces(env)
  for name, space in env.act_space.items():
    if not space.discrete:
      env = W.ClipAction(env, name)
  for k, v in _space_record(env.obs_space).items():
    out[f'obs_space/{k}'] = v
  for k, v in _space_record(env.act_space).items():
    out[f'act_space/{k}'] = v
  rng = np.random.default_rng(11)
  seen = []
  for t in range(12):
    action = {
        'action': rng.uniform(-1.6, 1.6, 2).astype(np.float32),   # beyond [-1, 1]: clipped
        'free': rng.uniform(-3, 3, 2).astype(np.float32),         # unbounded dims pass... then clip
        'choice': np.int32(rng.integers(0, 4)),
        'reset': bool(t == 0 or t == 9),
    }
    seen.append(env.step(action))
  for key in seen[0]:
    out[f'obs/{key}'] = np.stack([np.asarray(o[key]) for o in seen])
    out[f'obs_dtype/{key}'] = np.frombuffer(
        np.asarray(seen[-1][key]).dtype.str.encode().ljust(4), np.uint8).copy()
  for key in base.received[0]:
    out[f'inner/{key}'] = np.stack([r[key] for r in base.received])
    out[f'inner_dtype/{key}'] = np.frombuffer(
        base.received[-1][key].dtype.str.encode().ljust(4), np.uint8).copy()
  # Malformed actions at the CheckSpaces level (below ClipAction: address it directly).
  checker = W.CheckSpaces(W.UnifyDtypes(_ScriptedEnv(ns.Space)))
  good = {'action': np.zeros(2, np.float32), 'free': np.zeros(2, np.float32),
          'choice': np.int32(1), 'reset': True}
  bad = [
      {**good, 'action': np.zeros(3, np.float32)},            # shape
      {**good, 'action': np.array([5.0, 0.0], np.float32)},   # range
      {**good, 'choice': np.float32(1.5)},                    # dtype kind
      {**good, 'choice': 'one'},                              # foreign type
      good,
  ]
  codes = []
  for action in bad:
    try:
      checker.step(dict(action))
      codes.append(0)
    except TypeError:
      codes.append(1)
    except ValueError:
      codes.append(2)
  out['check_codes'] = np.array(codes)
  # TimeLimit (soft and hard) around ActionRepeat.
  for hard in (True, False):
    inner = _ScriptedEnv(ns.Space, episode=100)
    env = W.TimeLimit(W.ActionRepeat(inner, 3), duration=4, reset=hard)
    rows = []
    for t in range(14):
      obs = env.step({'action': np.zeros(2), 'free': np.zeros(2, np.float32),
                      'choice': np.int64(0), 'reset': t == 0})
      rows.append([obs['count'], obs['reward'], obs['is_first'], obs['is_last']])
    tag = 'hard' if hard else 'soft'
    out[f'limit_{tag}/rows'] = np.array(rows, np.float64)
    out[f'limit_{tag}/inner_resets'] = np.array([bool(r['reset']) for r in inner.received])
  try:
    env.no_such_attribute
    out['missing_attr'] = np.array(0)
  except ValueError:
    out['missing_attr'] = np.array(2)
  except AttributeError:
    out['missing_attr'] = np.array(1)
  return out


def sel_recency(ns):
  """Recency (selectors.py:60-125) through its public surface: the draws of a
  filling buffer (fewer items than table entries: the age is scaled), of a full
  one under FIFO eviction, a three-level table, and a flat prefix of zeros at
  the old end.  Reference side: its own class with the one-token repair of
  `_sample` (tests/adapters.py)."""
  out = {}
  # two levels (40 -> 16^2), the table of ppo/main.py:199 with recexp 1
  r = ns.Recency(1.0 / np.arange(1, 41) ** 1.0, seed=3)
  draws = []
  for k in range(12):
    r[k] = None
    draws += [r(), r()]
  out['filling'] = np.array(draws)
  live, draws = list(range(12)), []
  for k in range(12, 150):
    if len(live) >= 40:
      del r[live.pop(0)]
    r[k] = None
    live.append(k)
    draws.append(r())
  out['churn'] = np.array(draws)
  # three levels (300 -> 16^3), steeper table, full from the start
  r = ns.Recency(1.0 / np.arange(1, 301) ** 1.7, seed=11)
  for k in range(300):
    r[1000 + k] = None
  out['deep'] = np.array([r() for _ in range(200)])
  live = list(range(1000, 1300))
  draws = []
  for k in range(1300, 1500):
    del r[live.pop(0)]
    r[k] = None
    live.append(k)
    draws.append(r())
  out['deep_churn'] = np.array(draws)
  # a table whose old end is zero: those ages are never drawn
  table = np.concatenate([np.linspace(1.0, 0.1, 20), np.zeros(12)])
  r = ns.Recency(table, seed=5)
  for k in range(32):
    r[k] = None
  out['zero_tail'] = np.array([r() for _ in range(120)])
  out['lens'] = np.array([len(r)])
  return out


HOST_SCENARIOS = {
    'wrappers_chain': wrappers_chain,
    'sel_recency': sel_recency,
}
