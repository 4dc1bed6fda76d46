"""Context-only keys: `Replay(heads={'dyn/': K})` returns (batch, K, ...) for the
keys whose consumer reads only the head of the sampled window.  The reference's
DreamerV3 takes `x[:, :K]` (K = replay_context) of the sampled enc/ dyn/ dec/
entries (dreamerv3/agent.py:322-331) and its `_assemble_batch` copies a
`[start, stop)` sub-range of the window (embodied/core/replay.py:255-275).

Contract checked here, bit-exact: every key of a heads sample equals the
oracle's full sample sliced `[:, :K]` -- same draws, same PRNG stream, same
annotation -- and `update` still writes all T steps back."""
import numpy as np
import pytest
import torch

from oracle import np_oracle
from tests import scenarios
from tests.conftest import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def emb():
  import embodied_amd
  assert torch.cuda.is_available(), 'these tests need the MI355X'
  return embodied_amd


def _cut(want, heads, length):
  """The oracle's full batch cut to the heads (what the shipped agent's
  `lhs = x[:, :K]` would have read)."""
  def steps(name):
    if name in ('stepid', 'is_first', 'is_last'):
      return length
    best, k = -1, length
    for pattern, n in heads.items():
      hit = name == pattern or (pattern.endswith('/') and name.startswith(pattern))
      if hit and len(pattern) > best:
        best, k = len(pattern), n
    return k
  return {name: value[:, :steps(name)] for name, value in want.items()}


def _host(batch):
  return {k: v.cpu().numpy() for k, v in batch.items()}


@pytest.mark.parametrize('chunksize,L,K,batches', [
    (16, 7, 1, (1, 3, 16, 100)),   # windows cross chunk boundaries; 100 > the spans staged with the head
    (16, 7, 3, (5, 80)),           # a head that itself crosses the run split in some windows
    (64, 65, 1, (1, 16)),          # BASELINE length, K = replay_context = 1 (dreamerv3/configs.yaml:15)
    (8, 9, 2, (2, 40)),            # windows of 2-3 chunks: not span-shaped, row tables
    (16, 7, 7, (4,)),              # K = L: the plain sample
])
def test_heads_sample_equals_oracle_sample_sliced(emb, chunksize, L, K, batches):
  """Wide keys through the persistent span mover, narrow ones through the flat
  mover, mixed head lengths in one launch."""
  heads = {'dyn/': K, 'enc/feat': min(K + 1, L), 'small': K}
  n_workers, n_steps = 3, 6 * chunksize + L
  ours = emb.Replay(length=L, capacity=4 * n_steps, chunksize=chunksize, seed=5, heads=heads)
  ref = np_oracle.Replay(L, 4 * n_steps, chunksize, seed=5)
  gen = np.random.default_rng(1)

  def step(t, w):
    s = dict(scenarios.synth_step(t, w))
    s['dyn/deter'] = gen.standard_normal(520).astype(np.float32)       # 2080 B: wide, span mover
    s['dyn/stoch'] = gen.standard_normal((8, 16)).astype(np.float32)   # 512 B: 16-byte units, flat
    s['enc/feat'] = gen.integers(0, 255, (48, 48, 3), dtype=np.uint8)  # 6912 B: wide, its own head
    s['small'] = np.int16(t * 3 + w)                                   # 2-byte units
    s['full'] = gen.standard_normal(1028).astype(np.float16)           # not a head key: whole window
    return s

  for t in range(n_steps):
    for w in range(n_workers):
      s = step(t, w)
      ours.add(s, w)
      ref.add(s, w)
  for batch in batches:
    got = ours.sample(batch)
    assert got['dyn/deter'].shape == (batch, K, 520) and got['dyn/deter'].is_contiguous()
    assert got['full'].shape == (batch, L, 1028)
    assert_same(_host(got), _cut(ref.sample(batch), heads, L), f'sample {batch}')
  # write-back covers all T steps although only K came out (dreamerv3/agent.py:144-150)
  want = ref.sample(4)
  got = ours.sample(4)
  assert np.array_equal(got['stepid'].cpu().numpy(), want['stepid'])
  T = L - 1 if L > 1 else 1
  upd = {'stepid': want['stepid'][:, :T],
         'dyn/deter': gen.standard_normal((4, T, 520)).astype(np.float32),
         'dyn/stoch': gen.standard_normal((4, T, 8, 16)).astype(np.float32)}
  ours.update({k: torch.as_tensor(v).cuda() for k, v in upd.items()})
  ref.update(dict(upd))
  assert_same(_host(ours.sample(12)), _cut(ref.sample(12), heads, L), 'after update')
  # ... and what is in the pool is the full write-back, not just the head: a
  # whole-window gather of the next draw's rows equals the oracle's full sample
  state = ref.sample(6)
  rows, _ = ours.sample_index(6)
  assert_same(_host(ours.gather(rows)), state, 'pool after update')


@pytest.mark.parametrize('seed', range(5))
def test_random_schemas_with_heads_against_oracle(emb, seed):
  """Random key sets and dtypes (rows from 1 B to ~5 KB, aligned and odd), random
  head lengths on a random subset of keys, online and offline replays, pool
  growth, recycled and `out=` batches."""
  gen = np.random.default_rng(900 + seed)
  dtypes = [np.uint8, np.int16, np.int32, np.int64, np.float16, np.float32, np.float64, np.bool_]
  shapes = [(), (1,), (3,), (7,), (16,), (5, 3), (2, 4, 4), (640,), (33, 37), (1283,), (2048,)]
  spec = {}
  for i in range(int(gen.integers(2, 10))):
    prefix = ['dyn/', 'enc/', 'dec/', ''][int(gen.integers(0, 4))]
    spec[f'{prefix}k{i}'] = (dtypes[int(gen.integers(0, len(dtypes)))], shapes[int(gen.integers(0, len(shapes)))])
  length, chunksize = int(gen.integers(1, 9)), int(gen.integers(2, 40))
  heads = {}
  for pattern in ('dyn/', 'enc/', 'dec/'):
    if gen.random() < 0.7:
      heads[pattern] = int(gen.integers(1, length + 1))
  exact = [k for k in spec if '/' not in k]
  if exact and gen.random() < 0.5:
    heads[exact[0]] = int(gen.integers(1, length + 1))

  def value(dtype, shape, lead=()):
    full = (*lead, *shape)
    if dtype == np.bool_:
      return gen.random(full) < 0.5
    if np.issubdtype(dtype, np.integer):
      return gen.integers(0, 100, full).astype(dtype)
    return gen.standard_normal(full).astype(dtype)

  def step(t, lead=()):
    s = {k: value(d, sh, lead) for k, (d, sh) in spec.items()}
    s['is_first'] = np.full(lead, t % 9 == 0) if lead else np.bool_(t % 9 == 0)
    s['is_last'] = np.full(lead, t % 9 == 8) if lead else np.bool_(t % 9 == 8)
    return s

  workers = int(gen.integers(1, 5))
  capacity = int(gen.integers(5, 120))
  online = bool(gen.integers(0, 2))
  ours = emb.Replay(length, capacity, chunksize=chunksize, online=online, seed=seed,
                    stage_rows=int(gen.integers(1, 30)), slots=8, heads=heads)
  ref = np_oracle.Replay(length, capacity, chunksize, online, seed=seed)
  held = None
  for t in range(120):
    s = step(t, (workers,))
    ours.add_batch({k: torch.as_tensor(v).cuda() for k, v in s.items()}, list(range(workers)))
    for w in range(workers):
      ref.add({k: v[w] for k, v in s.items()}, w)
    if len(ref) and t % 7 == 0:
      B = int(gen.integers(1, 6))
      mode = ['train', 'report'][int(gen.integers(0, 2))]
      reuse = held is not None and held['stepid'].shape[0] == B and gen.random() < 0.5
      got = ours.sample(B, mode, out=held) if reuse else ours.sample(B, mode)
      want = _cut(ref.sample(B, mode), heads, length)
      assert_same(_host(got), want, f'seed{seed} t{t}')
      held = got
      if gen.random() < 0.5:
        T = int(gen.integers(1, length + 1))
        names = [k for k in spec if gen.random() < 0.5] or [next(iter(spec))]
        upd = {k: value(*spec[k], (B, T)) for k in names}
        ours.update({'stepid': got['stepid'][:, :T], **{k: torch.as_tensor(v).cuda() for k, v in upd.items()}})
        ref.update({'stepid': want['stepid'][:, :T], **upd})


def test_replay_context_as_the_shipped_agent_consumes_it(emb):
  """`_apply_replay_context` restated (dreamerv3/agent.py:312-331): from the
  sampled data it reads `entries[:, :K]` of enc/ dyn/ dec/ and `[:, K:]` of the
  observations, actions and step ids.  Run on a full sample and on a heads
  sample of the same draws, it must see identical inputs -- the shipped agent
  needs no change -- and its T-step write-back lands the same way."""
  L, T, K, B = 9, 8, 1, 6
  heads = {'enc/': K, 'dyn/': K, 'dec/': K}
  reps = [emb.Replay(L, 200, chunksize=16, seed=3),
          emb.Replay(L, 200, chunksize=16, seed=3, heads=heads)]
  gen = torch.Generator(device='cuda').manual_seed(0)
  for t in range(60):
    step = {
        'image': torch.randint(0, 255, (4, 8, 8, 3), dtype=torch.uint8, device='cuda', generator=gen),
        'reward': torch.randn(4, device='cuda', generator=gen),
        'is_first': torch.full((4,), t % 13 == 0, device='cuda'),
        'is_last': torch.full((4,), t % 13 == 12, device='cuda'),
        'is_terminal': torch.zeros(4, dtype=torch.bool, device='cuda'),
        'action': torch.randn((4, 6), device='cuda', generator=gen),
        'dyn/deter': torch.randn((4, 512), device='cuda', generator=gen),
        'dyn/stoch': torch.randn((4, 8, 16), device='cuda', generator=gen),
        'enc/none': torch.zeros((4, 1), device='cuda'),
    }
    for rep in reps:
      rep.add_batch(step, [0, 1, 2, 3])

  def apply_replay_context(data):
    entries = {k: v[:, :K] for k, v in data.items() if k.split('/')[0] in ('enc', 'dyn', 'dec')}
    obs = {k: data[k][:, K:] for k in ('image', 'reward', 'is_first', 'is_last', 'is_terminal')}
    prevact = data['action'][:, K - 1: -1]
    return entries, obs, prevact, data['stepid'][:, K:]

  for _ in range(3):
    full, cut = reps[0].sample(B), reps[1].sample(B)
    assert cut['dyn/deter'].shape == (B, K, 512) and full['dyn/deter'].shape == (B, L, 512)
    a, b = apply_replay_context(full), apply_replay_context(cut)
    for x, y in zip(a[:2], b[:2]):
      assert x.keys() == y.keys()
      for k in x:
        assert torch.equal(x[k], y[k]), k
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    # train's write-back: entries for the T trained steps, stepid = rhs(stepid)
    deter = torch.randn((B, T, 512), device='cuda', generator=gen)
    stoch = torch.randn((B, T, 8, 16), device='cuda', generator=gen)
    for rep, (_, _, _, stepid) in zip(reps, (a, b)):
      rep.update({'stepid': stepid, 'dyn/deter': deter, 'dyn/stoch': stoch})
  # both pools hold the same bytes afterwards: a full gather of the same rows agrees
  rows = reps[0].sample_index(B)[0]
  rows_b = reps[1].sample_index(B)[0]
  assert np.array_equal(rows, rows_b)
  x, y = reps[0].gather(rows), reps[1].gather(rows)
  for k in x:
    assert torch.equal(x[k], y[k]), k


def test_heads_at_the_dreamer_shapes(emb):
  """configs[2] shapes: 64 envs, 84x84x4 frames + 40 KB of f32 latents per step,
  L = 65, B = 16, K = 1: the latents come back (16, 1, ...), equal to the full
  gather's first step, everything else unchanged."""
  from embodied_amd.envs import synthetic
  n_env, L, B, cap = 64, 65, 16, 3000
  reps = [emb.Replay(L, cap, chunksize=1024, seed=0),
          emb.Replay(L, cap, chunksize=1024, seed=0, heads={'dyn/': 1})]
  env = synthetic.SyntheticBatchEnv(n_env, episode_len=100, ring=4)
  reset = torch.ones(n_env, dtype=torch.bool, device='cuda')
  gen = torch.Generator(device='cuda').manual_seed(2)
  for tick in range(cap // n_env + 2 * L):
    obs = env.step({'reset': reset})
    reset = obs['is_last']
    step = {**obs, 'action': torch.zeros(n_env, dtype=torch.int32, device='cuda'),
            'dyn/deter': torch.randn((n_env, 8192), device='cuda', generator=gen),
            'dyn/stoch': torch.randn((n_env, 32, 64), device='cuda', generator=gen)}
    for rep in reps:
      rep.add_batch(step, list(range(n_env)))
  for _ in range(4):
    full, cut = reps[0].sample(B), reps[1].sample(B)
    for k in full:
      n = 1 if k.startswith('dyn/') else L
      assert cut[k].shape[:2] == (B, n), k
      assert torch.equal(cut[k], full[k][:, :n]), k


def test_heads_are_checked(emb):
  with pytest.raises(ValueError):
    emb.Replay(5, 10, heads={'dyn/': 6})
  with pytest.raises(ValueError):
    emb.Replay(5, 10, heads={'dyn/': 0})
  rep = emb.Replay(4, 50, chunksize=8, seed=0, heads={'x': 2, 'is_first': 1, 'stepid': 1})
  for t in range(20):
    rep.add({'x': np.float32(t), 'y': np.int32(t), 'is_first': np.bool_(t == 0), 'is_last': np.bool_(False)})
  got = rep.sample(3)
  # the flags and step ids stay whole whatever `heads` says: update and the
  # episode bookkeeping of every consumer need them per step
  assert got['x'].shape == (3, 2) and got['y'].shape == (3, 4)
  assert got['is_first'].shape == (3, 4) and got['stepid'].shape == (3, 4, 20)
  wrong = {k: torch.empty_like(v) for k, v in rep.sample(3).items()}
  wrong['x'] = torch.empty((3, 4), dtype=torch.float32, device='cuda')
  with pytest.raises(ValueError):
    rep.sample(3, out=wrong)


def test_windowed_routes_refuse_context_only_keys(emb):
  """A head holds for the whole sampled sequence, not for each `Consec` window:
  both windowed routes (the fused gather and the slicing one) say so instead of
  handing back keys of two different time axes."""
  rep = emb.Replay(8, 100, chunksize=8, seed=0, heads={'dyn/': 1})
  for t in range(30):
    rep.add({'image': np.uint8(t), 'dyn/deter': np.float32(t), 'is_first': np.bool_(t == 0),
             'is_last': np.bool_(False)})
  with pytest.raises(ValueError, match='heads='):
    rep.sample_windows(2, 4, 2)
  windows = emb.streams.Consec(emb.streams.Stateless(rep.sample, 2), length=4, consec=2)
  with pytest.raises(AssertionError, match='context-only'):
    next(iter(windows))
  # one window per sequence is the sequence itself: heads pass through
  whole = next(iter(emb.streams.Consec(emb.streams.Stateless(rep.sample, 2), length=8, consec=1)))
  assert whole['dyn/deter'].shape == (2, 1) and whole['image'].shape == (2, 8)
  # ... and a single window that is NOT the whole sequence (strict=False, more steps
  # than needed) is refused like the others, whichever key comes first in the dict
  cut = emb.streams.Consec(emb.streams.Stateless(rep.sample, 2), length=6, consec=1, strict=False)
  with pytest.raises(AssertionError, match='context-only'):
    next(iter(cut))
  batch = rep.sample(2)
  for order in (list(batch), list(batch)[::-1]):
    with pytest.raises(ValueError, match='cannot be windowed'):
      emb.streams.window_batch({k: batch[k] for k in order}, 1, 4)
  full = {k: v for k, v in batch.items() if v.shape[1] == 8}
  got = emb.streams.window_batch(full, 1, 4)
  assert all(torch.equal(got[k], full[k][:, 1:5]) for k in full)
