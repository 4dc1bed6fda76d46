"""CPU oracle: a numpy restatement of the reference's Driver / Replay hot path.

TEST INFRASTRUCTURE — NOT THE PRODUCT.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import this module; the product
package `embodied_amd` never does (it fails loudly without its HIP library).

Every class restates, in plain sequential numpy, what the reference computes for
one piece of the path, citing the reference file:line it follows (paths relative
to `/root/reference/`).  It is deliberately single-threaded and lock-free: it is
the *checker*, so only values matter.

Parity pinning (see tests/test_oracle_golden.py, oracle/gen_golden.py):
  * selectors / Replay / Driver / Consec are pinned against golden vectors
    produced by running the real reference modules in the build container
    (`tests/golden/*.npz`, generator committed).
  * The return scans (GAE, lambda-return, Director score, split/abstract_traj)
    live in JAX code that cannot be imported here (no jax).  They are pinned
    against fixtures made by EXECUTING the reference's own function text under
    numpy stand-ins for `jnp`/`f32`/`sg`/`chex` (`oracle/gen_scan_golden.py`,
    `oracle/shims/jaxlike.py` -> `tests/golden/scan_*.npz`): this oracle equals
    them bit for bit in float32 (`tests/test_scan_golden.py`).  numpy stands in
    for XLA's float32 arithmetic there (FMA contraction may differ at ~1e-7).

Integer PRNG: the reference draws through `numpy.random.default_rng`; so does
this oracle (numpy is the actual third-party dependency).  The product restates
PCG64 in C++ and is compared with this oracle bit for bit.
"""
import collections

import numpy as np


# ----------------------------------------------------------------------------
# Selectors   (embodied/core/selectors.py)
# ----------------------------------------------------------------------------


class Fifo:
  """Oldest key first (selectors.py:7-26)."""

  def __init__(self):
    self.order = collections.deque()

  def __call__(self):
    return self.order[0]

  def __len__(self):
    return len(self.order)

  def __setitem__(self, key, stepids):
    self.order.append(key)

  def __delitem__(self, key):
    if self.order[0] == key:
      self.order.popleft()
    else:
      self.order.remove(key)


class Uniform:
  """Dense key table with swap-remove; one bounded integer draw per sample
  (selectors.py:29-57).  `allow_single` relaxes the reference's `2 <= len`
  assertion on delete (selectors.py:52) that makes a capacity-1 replay raise;
  see DESIGN.md "Known reference defects"."""

  def __init__(self, seed=0, allow_single=True):
    self.rng = np.random.default_rng(seed)
    self.table = []
    self.where = {}
    self.allow_single = allow_single

  def __len__(self):
    return len(self.table)

  def __call__(self):
    pos = self.rng.integers(0, len(self.table)).item()
    return self.table[pos]

  def __setitem__(self, key, stepids):
    self.where[key] = len(self.table)
    self.table.append(key)

  def __delitem__(self, key):
    if not self.allow_single:
      assert len(self.table) >= 2
    pos = self.where.pop(key)
    tail = self.table.pop()
    if pos != len(self.table):
      self.table[pos] = tail
      self.where[tail] = pos


class _Inner:
  __slots__ = ('up', 'kids', 'mass')

  def __init__(self):
    self.up = None
    self.kids = []
    self.mass = 0


class _Leaf:
  __slots__ = ('up', 'key', 'mass')

  def __init__(self, key, mass):
    self.up = None
    self.key = key
    self.mass = mass


def _resum(node):
  # selectors.py:340-344: every ancestor recomputes its mass from scratch as a
  # left-to-right Python `sum` (plain float adds on CPython <= 3.11).
  while node is not None:
    total = 0
    for kid in node.kids:
      total = total + kid.mass
    node.mass = total
    node = node.up


def _attach(parent, child):
  # selectors.py:327-332
  if child.up is not None:
    _detach(child.up, child)
  child.up = parent
  parent.kids.append(child)
  _resum(parent)


def _detach(parent, child):
  # selectors.py:334-338
  child.up = None
  parent.kids.remove(child)
  _resum(parent)


class SampleTree:
  """b-ary sum tree with append-at-tail insertion and move-tail-into-hole
  removal (selectors.py:231-306)."""

  def __init__(self, branching=16, seed=0):
    assert branching >= 2
    self.branching = branching
    self.root = _Inner()
    self.tail = None
    self.leaves = {}
    self.rng = np.random.default_rng(seed)

  def __len__(self):
    return len(self.leaves)

  def insert(self, key, mass):
    # selectors.py:244-264
    if self.tail is None:
      spot = self.root
    else:
      climbed = 0
      spot = self.tail.up
      while spot is not None and len(spot.kids) >= self.branching:
        spot = spot.up
        climbed += 1
      if spot is None:
        spot = _Inner()
        _attach(spot, self.root)
        self.root = spot
      for _ in range(climbed):
        fresh = _Inner()
        _attach(spot, fresh)
        spot = fresh
    leaf = _Leaf(key, mass)
    _attach(spot, leaf)
    self.leaves[key] = leaf
    self.tail = leaf

  def remove(self, key):
    # selectors.py:266-285
    leaf = self.leaves.pop(key)
    hole_parent = leaf.up
    tail_parent = self.tail.up
    _detach(hole_parent, leaf)
    if leaf is not self.tail:
      _attach(hole_parent, self.tail)
    node = tail_parent
    while node.up is not None and not node.kids:
      above = node.up
      _detach(above, node)
      node = above
    if not node.kids:
      self.tail = None
      return
    while isinstance(node, _Inner):
      node = node.kids[-1]
    self.tail = node

  def update(self, key, mass):
    # selectors.py:287-290
    leaf = self.leaves[key]
    leaf.mass = mass
    _resum(leaf.up)

  def sample(self):
    # selectors.py:292-306
    node = self.root
    while isinstance(node, _Inner):
      masses = np.array([kid.mass for kid in node.kids])
      total = masses.sum()
      if not np.isfinite(total):
        hot = np.isinf(masses)
        probs = hot / hot.sum()
      elif total == 0:
        probs = np.ones(len(masses)) / len(masses)
      else:
        probs = masses / total
      pick = self.rng.choice(np.arange(len(masses)), p=probs)
      node = node.kids[pick.item()]
    return node.key

  def layout(self):
    """Nested (key, mass) lists in child order: lets tests compare tree shape
    and float sums with the C++ core exactly."""
    def walk(node):
      if isinstance(node, _Leaf):
        return (node.key, float(node.mass))
      return [walk(kid) for kid in node.kids]
    return walk(self.root)


class Prioritized:
  """Per-step priorities aggregated into per-item masses on a SampleTree
  (selectors.py:128-197).  Priorities are taken as float64 (what the reference
  computes under its pinned numpy<2 scalar promotion)."""

  def __init__(self, exponent=1.0, initial=1.0, zero_on_sample=False,
               maxfrac=0.0, branching=16, seed=0):
    assert 0 <= maxfrac <= 1
    self.exponent = float(exponent)
    self.initial = float(initial)
    self.zero_on_sample = zero_on_sample
    self.maxfrac = maxfrac
    self.tree = SampleTree(branching, seed)
    self.prio = {}      # stepid bytes -> float
    self.users = {}     # stepid bytes -> [item keys]
    self.items = {}     # item key -> [stepid bytes]

  @staticmethod
  def _asbytes(stepids):
    if len(stepids) and not isinstance(stepids[0], bytes):
      return [np.asarray(x, np.uint8).tobytes() for x in stepids]
    return list(stepids)

  def __len__(self):
    return len(self.items)

  def __call__(self):
    # selectors.py:163-168
    key = self.tree.sample()
    if self.zero_on_sample:
      self.prioritize(self.items[key], [0.0] * len(self.items[key]))
    return key

  def __setitem__(self, key, stepids):
    # selectors.py:170-175
    stepids = self._asbytes(stepids)
    self.items[key] = stepids
    for sid in stepids:
      self.users.setdefault(sid, []).append(key)
      self.prio.setdefault(sid, self.initial)
    self.tree.insert(key, self._mass(key))

  def __delitem__(self, key):
    # selectors.py:177-185
    self.tree.remove(key)
    for sid in self.items.pop(key):
      group = self.users[sid]
      group.remove(key)
      if not group:
        del self.users[sid]
        del self.prio[sid]

  def prioritize(self, stepids, priorities):
    # selectors.py:143-158.  The table takes a priority for ANY step id (`prios`
    # is a defaultdict, :139,147): one that belongs to no item yet keeps it, and
    # an item that arrives later aggregates it (__setitem__ -> _mass).
    stepids = self._asbytes(stepids)
    touched = []
    for sid, value in zip(stepids, priorities):
      self.prio[sid] = float(value)
      touched += self.users.get(sid, [])
    for key in set(touched):
      self.tree.update(key, self._mass(key))

  def _mass(self, key):
    # selectors.py:187-197
    vals = [self.prio[sid] for sid in self.items[key]]
    if self.exponent != 1.0:
      vals = [v ** self.exponent for v in vals]
    total = 0
    for v in vals:
      total = total + v
    mean = total / len(vals)
    if self.maxfrac:
      return self.maxfrac * max(vals) + (1 - self.maxfrac) * mean
    return mean


class Mixture:
  """Pick a member selector by probability, then delegate
  (selectors.py:200-228)."""

  def __init__(self, selectors, fractions, seed=0):
    assert set(selectors) == set(fractions)
    assert sum(fractions.values()) == 1
    names = sorted(k for k in selectors if fractions[k])
    self.members = [selectors[k] for k in names]
    self.fractions = np.array([fractions[k] for k in names], np.float32)
    self.rng = np.random.default_rng(seed)

  def __call__(self):
    pick = self.rng.choice(len(self.members), p=self.fractions)
    return self.members[int(pick)]()

  def __len__(self):
    return len(self.members[0])

  def __setitem__(self, key, stepids):
    for member in self.members:
      member[key] = stepids

  def __delitem__(self, key):
    for member in self.members:
      del member[key]

  def prioritize(self, stepids, priorities):
    for member in self.members:
      if hasattr(member, 'prioritize'):
        member.prioritize(stepids, priorities)


# ----------------------------------------------------------------------------
# Replay   (embodied/core/replay.py, embodied/core/chunk.py)
# ----------------------------------------------------------------------------


def make_stepid(uid, index):
  """20-byte step id = 16-byte big-endian chunk uid || 4-byte big-endian row
  (replay.py:90-91).  The reference's uid is an `elements.UUID`; byte content
  is unpinned, the build uses a per-replay chunk serial."""
  return np.frombuffer(
      int(uid).to_bytes(16, 'big') + int(index).to_bytes(4, 'big'), np.uint8)


class _Block:
  """Fixed-size time-major column store (chunk.py:9-62)."""

  __slots__ = ('uid', 'succ', 'fill', 'size', 'cols')

  def __init__(self, uid, size):
    self.uid = uid
    self.succ = 0
    self.fill = 0
    self.size = size
    self.cols = None

  def append(self, step):
    # chunk.py:41-50
    assert self.fill < self.size
    if self.cols is None:
      self.cols = {
          k: np.empty((self.size, *v.shape), v.dtype) for k, v in step.items()}
    for k, v in step.items():
      self.cols[k][self.fill] = v
    self.fill += 1

  def window(self, index, count):
    # chunk.py:60-62
    assert 0 <= index and index + count <= self.fill
    return {k: v[index: index + count] for k, v in self.cols.items()}

  def write(self, index, count, values):
    # chunk.py:54-58
    assert 0 <= index and index + count <= self.fill
    for k, v in values.items():
      self.cols[k][index: index + count] = v

  @property
  def nbytes(self):
    return sum(x.nbytes for x in self.cols.values()) if self.cols else 0


class Replay:
  """Chunked sequence replay (replay.py:14-292, 362-370)."""

  def __init__(self, length, capacity=None, chunksize=1024, online=False,
               selector=None, seed=0):
    self.length = length
    self.capacity = capacity and int(capacity)
    self.chunksize = chunksize
    self.sampler = selector if selector is not None else Uniform(seed)
    self.blocks = {}
    self.refs = {}
    self.items = {}
    self.fifo = collections.deque()
    self.next_item = 0
    self.next_uid = 1
    self.cursor = {}
    self.pending = collections.defaultdict(collections.deque)
    self.online = online
    self.steps_seen = collections.defaultdict(int)
    self.fresh = collections.deque()
    self.metrics = {'samples': 0, 'inserts': 0, 'updates': 0}

  def __len__(self):
    return len(self.items)

  def _new_block(self, refs):
    block = _Block(self.next_uid, self.chunksize)
    self.next_uid += 1
    self.blocks[block.uid] = block
    self.refs[block.uid] = refs
    return block

  def add(self, step, worker=0):
    # replay.py:77-118
    step = {k: np.asarray(v) for k, v in step.items()
            if not k.startswith('log/')}
    if worker not in self.cursor:
      self.cursor[worker] = (self._new_block(1).uid, 0)
    uid, index = self.cursor[worker]
    step['stepid'] = make_stepid(uid, index)
    block = self.blocks[uid]
    block.append(step)
    queue = self.pending[worker]
    queue.append((uid, index))
    self.refs[uid] += 1
    index += 1
    if index < block.size:
      self.cursor[worker] = (uid, index)
    else:
      self._rotate(block, worker)
    if len(queue) >= self.length:
      self.metrics['inserts'] += 1
      uid, index = queue.popleft()
      self._insert(uid, index)
      if self.online and self.steps_seen[worker] % self.length == 0:
        self.fresh.append((uid, index))
    if self.online:
      self.steps_seen[worker] += 1

  def _rotate(self, block, worker):
    # replay.py:362-370
    succ = self._new_block(2)
    self.refs[block.uid] -= 1
    self.cursor[worker] = (succ.uid, 0)
    block.succ = succ.uid

  def _insert(self, uid, index):
    # replay.py:171-179
    while self.capacity and len(self.items) >= self.capacity:
      self._evict()
    key = self.next_item
    self.next_item += 1
    self.items[key] = (uid, index)
    self.sampler[key] = self.rows(uid, index, self.length, ['stepid'])['stepid']
    self.fifo.append(key)

  def _evict(self):
    # replay.py:181-191
    key = self.fifo.popleft()
    del self.sampler[key]
    uid, _ = self.items.pop(key)
    self.refs[uid] -= 1
    if self.refs[uid] < 1:
      del self.refs[uid]
      block = self.blocks.pop(uid)
      if block.succ in self.refs:
        self.refs[block.succ] -= 1

  def spans(self, uid, index, count):
    """[(block uid, first row, rows)] covering `count` steps (replay.py:193-214).
    Raises KeyError if the first block was evicted."""
    block = self.blocks[uid]
    have = block.fill - index
    if have >= count:
      return [(uid, index, count)]
    out = [(uid, index, have)]
    left = count - have
    while left > 0:
      block = self.blocks[block.succ]
      used = min(left, block.fill)
      out.append((block.uid, 0, used))
      left -= used
    return out

  def rows(self, uid, index, count, keys=None):
    parts = [self.blocks[u].window(i, n) for u, i, n in self.spans(uid, index, count)]
    keys = keys or parts[0].keys()
    return {k: np.concatenate([p[k] for p in parts], 0) for k in keys}

  def draw(self, mode='train'):
    """One sequence start: (block uid, row, came_from_online_queue)
    (replay.py:151-169)."""
    assert mode in ('train', 'report', 'eval')
    if mode == 'train':
      self.metrics['samples'] += 1
    while True:
      if self.online and self.fresh and mode == 'train':
        uid, index = self.fresh.popleft()
        online = True
      else:
        uid, index = self.items[self.sampler()]
        online = False
      if uid in self.blocks:
        return uid, index, online

  def sample(self, batch, mode='train'):
    # replay.py:121-127, 255-292
    assert len(self.sampler), 'oracle does not block on an empty buffer'
    starts = [self.draw(mode) for _ in range(batch)]
    seqs = [self.rows(uid, index, self.length) for uid, index, _ in starts]
    data = {k: np.stack([s[k] for s in seqs]) for k in seqs[0]}
    return annotate(data)

  def sample_starts(self, batch, mode='train'):
    """Index-only form used by parity tests: list of span lists."""
    starts = [self.draw(mode) for _ in range(batch)]
    return [self.spans(uid, index, self.length) for uid, index, _ in starts]

  def update(self, data):
    # replay.py:129-149, 216-235
    data = dict(data)
    stepid = np.asarray(data.pop('stepid'))
    priority = data.pop('priority', None)
    assert stepid.ndim == 3
    self.metrics['updates'] += int(np.prod(stepid.shape[:-1]))
    if priority is not None:
      assert np.ndim(priority) == 2
      self.sampler.prioritize(
          stepid.reshape((-1, stepid.shape[-1])),
          np.asarray(priority, np.float64).flatten())
    if not data:
      return
    for row in range(len(stepid)):
      raw = stepid[row, 0].tobytes()
      uid = int.from_bytes(raw[:16], 'big')
      index = int.from_bytes(raw[16:], 'big')
      count = len(next(iter(data.values()))[row])
      try:
        spans = self.spans(uid, index, count)
      except KeyError:
        continue
      at = 0
      for u, i, n in spans:
        self.blocks[u].write(
            i, n, {k: np.asarray(v[row])[at: at + n] for k, v in data.items()})
        at += n

  def stats(self):
    # replay.py:58-74
    m = self.metrics
    out = {
        'items': len(self.items),
        'chunks': len(self.blocks),
        'streams': len(self.pending),
        'ram_gb': sum(b.nbytes for b in self.blocks.values()) / 1024 ** 3,
        'inserts': m['inserts'],
        'samples': m['samples'],
        'updates': m['updates'],
        'replay_ratio': (
            self.length * m['samples'] / m['inserts'] if m['inserts']
            else np.nan),
    }
    self.metrics = {k: 0 for k in m}
    return out


def annotate(data):
  """is_first[:,0]=1; is_last |= next step's is_first (replay.py:277-292)."""
  data = dict(data)
  if 'is_first' in data:
    first = data['is_first'].copy()
    first[:, 0] = True
    data['is_first'] = first
    if 'is_last' in data:
      nxt = np.zeros_like(first)
      nxt[:, :-1] = first[:, 1:]
      data['is_last'] = data['is_last'] | nxt
  return data


# ----------------------------------------------------------------------------
# Streams   (embodied/core/streams.py:12-29, 89-150)
# ----------------------------------------------------------------------------


class Consec:
  """Serve one long sampled batch as `consec` overlapping windows
  (streams.py:89-150)."""

  def __init__(self, source, length, consec, prefix=0, strict=True):
    self.source = source
    self.length = length
    self.consec = consec
    self.prefix = prefix
    self.strict = strict
    self.turn = 0
    self.batch = None

  def __iter__(self):
    return self

  def __next__(self):
    if self.turn >= self.consec:
      self.turn = 0
    if self.turn == 0:
      self.batch = self.source()
      have = self.batch['is_first'].shape[1]
      need = self.length * self.consec + self.prefix
      assert need <= have
      if self.strict:
        assert need == have
    lo = self.turn * self.length
    hi = lo + self.length + self.prefix
    out = {k: np.ascontiguousarray(v[:, lo:hi]) for k, v in self.batch.items()}
    out['consec'] = np.full(out['is_first'].shape, self.turn, np.int32)
    self.turn += 1
    return out


# ----------------------------------------------------------------------------
# Driver   (embodied/core/driver.py:34-87)
# ----------------------------------------------------------------------------


class Driver:
  """Serial vectorised env loop (driver.py:11-87, parallel=False branch)."""

  def __init__(self, envs):
    self.envs = list(envs)
    self.n = len(self.envs)
    self.act_space = self.envs[0].act_space
    self.callbacks = []
    self.reset()

  def reset(self, init_policy=None):
    self.acts = {
        k: np.zeros((self.n,) + tuple(v.shape), v.dtype)
        for k, v in self.act_space.items()}
    self.acts['reset'] = np.ones(self.n, bool)
    self.carry = init_policy and init_policy(self.n)

  def on_step(self, fn):
    self.callbacks.append(fn)

  def __call__(self, policy, steps=0, episodes=0):
    step = episode = 0
    while step < steps or episode < episodes:
      step, episode = self.step(policy, step, episode)

  def step(self, policy, step=0, episode=0):
    per_env = [{k: v[i] for k, v in self.acts.items()} for i in range(self.n)]
    obs = [env.step(act) for env, act in zip(self.envs, per_env)]
    obs = {k: np.stack([o[k] for o in obs]) for k in obs[0]}
    logs = {k: v for k, v in obs.items() if k.startswith('log/')}
    obs = {k: v for k, v in obs.items() if not k.startswith('log/')}
    self.carry, acts, outs = policy(self.carry, obs)
    assert not set(acts) & set(outs)
    done = obs['is_last']
    if done.any():
      acts = {k: mask_rows(v, ~done) for k, v in acts.items()}
    self.acts = {**acts, 'reset': done.copy()}
    trans = {**obs, **acts, **outs, **logs}
    for i in range(self.n):
      tran = {k: v[i] for k, v in trans.items()}
      for fn in self.callbacks:
        fn(tran, i)
    return step + self.n, episode + int(done.sum())


def mask_rows(value, keep):
  # driver.py:84-87
  keep = keep.reshape(keep.shape + (1,) * (value.ndim - keep.ndim))
  return value * keep.astype(value.dtype)


def stack_obs(per_env_obs):
  """driver.py:65 — the N x S byte copy."""
  return {k: np.stack([o[k] for o in per_env_obs]) for k in per_env_obs[0]}


# ----------------------------------------------------------------------------
# Return scans (float32, sequential from the last step backwards).
# The reference code is JAX; pinned by executing its source under numpy
# stand-ins (oracle/gen_scan_golden.py, tests/test_scan_golden.py).
# ----------------------------------------------------------------------------

f32 = np.float32


def gae(rew, val, last, term, hor=200, lam=0.8):
  """PPO generalised advantage estimate (ppo/agent.py:188-201).
  rew,val f32 (B,T); last,term bool (B,T) -> adv,tar f32 (B,T-1)."""
  rew, val = rew.astype(f32), val.astype(f32)
  live = (~term).astype(f32)[:, 1:] * f32(1 - 1 / hor)
  cont = (~last & ~term).astype(f32)[:, 1:] * f32(lam)
  delta = rew[:, 1:] + live * val[:, 1:] - val[:, :-1]
  adv = np.zeros_like(delta)
  nxt = np.zeros(len(rew), f32)
  for t in reversed(range(delta.shape[1])):
    nxt = delta[:, t] + live[:, t] * cont[:, t] * nxt
    adv[:, t] = nxt
  return adv, adv + val[:, :-1]


def lambda_return(last, term, rew, boot, disc=1.0, lam=0.95):
  """DreamerV3 lambda-return (dreamerv3/agent.py:482-490) -> (B,T-1) f32."""
  rew, boot = rew.astype(f32), boot.astype(f32)
  live = (f32(1) - term.astype(f32))[:, 1:] * f32(disc)
  cont = (f32(1) - last.astype(f32))[:, 1:] * f32(lam)
  interm = rew[:, 1:] + (f32(1) - cont) * live * boot[:, 1:]
  ret = np.zeros_like(interm)
  nxt = boot[:, -1].copy()
  for t in reversed(range(interm.shape[1])):
    nxt = interm[:, t] + live[:, t] * cont[:, t] * nxt
    ret[:, t] = nxt
  return ret


def director_score(rew, cont, value, horizon=333, lam=0.95):
  """Director critic target, time-major (director/agent.py:430-445).
  rew (T-1,B), cont,value (T,B) -> ret (T-1,B) f32."""
  rew, cont, value = rew.astype(f32), cont.astype(f32), value.astype(f32)
  disc = cont[1:] * f32(1 - 1 / horizon)
  interm = rew + disc * value[1:] * f32(1 - lam)
  ret = np.zeros_like(interm)
  nxt = value[-1].copy()
  for t in reversed(range(len(disc))):
    nxt = interm[t] + disc[t] * f32(lam) * nxt
    ret[t] = nxt
  return ret


def split_traj(x, k, is_reward=False):
  """Director worker windows (director/hierarchy.py:224-238): time-major
  (T,B,...) -> (k, (T/k)*B, ...); reward keys are shifted by one step."""
  if is_reward:
    x = np.concatenate([0 * x[:1], x], 0)
  x = x.reshape((x.shape[0] // k, k) + x.shape[1:])
  x = np.moveaxis(x, 0, 1)
  x = x.reshape((x.shape[0], -1) + x.shape[3:])
  return x[1:] if is_reward else x


def abstract_traj(x, cont, k, kind='first'):
  """Director manager steps (director/hierarchy.py:240-256).  kind is one of
  'reward' (cumprod(cont)-weighted mean, shifted), 'cont' (product), 'first'."""
  fold = lambda a: a.reshape((a.shape[0] // k, k) + a.shape[1:])
  if kind == 'reward':
    w = np.cumprod(fold(cont), 1)
    x = np.concatenate([0 * x[:1], x], 0)
    return (fold(x) * w).mean(1)[1:]
  if kind == 'cont':
    return fold(x).prod(1)
  return fold(x)[:, 0]


def scan_closed_form(a, b, seed):
  """Independent float64 check of y_t = a_t + b_t * y_{t+1} along axis 1."""
  a, b = a.astype(np.float64), b.astype(np.float64)
  out = np.zeros_like(a)
  for i in range(a.shape[0]):
    for t in range(a.shape[1]):
      acc, w = 0.0, 1.0
      for s in range(t, a.shape[1]):
        acc += w * a[i, s]
        w *= b[i, s]
      out[i, t] = acc + w * float(seed[i])
  return out


# ----------------------------------------------------------------------------
# Normaliser statistics over data-parallel ranks (embodied/jax/utils.py:16-88).
# The class is a ninjax module (JAX absent) and the reference has no test for
# it.  Pinned since round 4 by EXECUTING the reference's class under numpy
# stand-ins (oracle/gen_normalize_golden.py -> tests/golden/normalize.npz): this
# restatement returns bit-identical statistics over five configurations x 40
# steps (tests/test_normalize_golden.py).  It also serves as the single-process
# statement of what the ranks must agree on.
# ----------------------------------------------------------------------------


class Normalize:
  """`update(parts)` takes the list of every rank's values of one step:
  Normalize._mean = local mean then pmean over ranks (utils.py:76-81),
  Normalize._perc = percentile of the all-gathered values (utils.py:83-88)."""

  def __init__(self, impl, rate=0.01, limit=1e-8, perclo=5.0, perchi=95.0, debias=True):
    assert impl in ('none', 'meanstd', 'perc'), impl
    self.impl, self.rate, self.limit = impl, f32(rate), f32(limit)
    self.perclo, self.perchi, self.debias = perclo, perchi, debias
    self.var = {k: f32(0) for k in ('corr', 'mean', 'sqrs', 'lo', 'hi')}

  def _update(self, name, x):                       # utils.py:90-91
    self.var[name] = f32((f32(1) - self.rate) * self.var[name] + self.rate * f32(x))

  def update(self, parts):                          # utils.py:44-57
    parts = [np.asarray(p, np.float32) for p in parts]
    if self.impl == 'meanstd':
      self._update('mean', np.mean([p.mean(dtype=np.float32) for p in parts], dtype=np.float32))
      self._update('sqrs', np.mean([np.square(p).mean(dtype=np.float32) for p in parts],
                                   dtype=np.float32))
    elif self.impl == 'perc':
      together = np.concatenate([p.reshape(-1) for p in parts])
      self._update('lo', np.percentile(together, self.perclo).astype(np.float32))
      self._update('hi', np.percentile(together, self.perchi).astype(np.float32))
    if self.debias and self.impl != 'none':
      self._update('corr', 1.0)

  def stats(self):                                  # utils.py:59-74
    if self.impl == 'none':
      return f32(0), f32(1)
    corr = f32(1)
    if self.debias:
      corr = f32(corr / np.maximum(self.rate, self.var['corr']))
    if self.impl == 'meanstd':
      mean = f32(self.var['mean'] * corr)
      std = np.sqrt(np.maximum(f32(0), f32(self.var['sqrs'] * corr - mean ** 2)))
      return mean, np.maximum(self.limit, std)
    lo, hi = f32(self.var['lo'] * corr), f32(self.var['hi'] * corr)
    return lo, np.maximum(self.limit, f32(hi - lo))
