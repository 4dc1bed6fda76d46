"""BASELINE.json's full sizes through size-independent properties (the oracle
cannot hold 10^5..10^6-step buffers of 28 KB frames in seconds): every gathered
byte is the generator's byte for the (env, step) the row claims to be, windows
are consecutive steps of one env, eviction is FIFO, write-backs read back."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def emb():
  import embodied_amd
  return embodied_amd


def _fill(emb, rep, n_env, steps, extra=None, episode_len=1000):
  from embodied_amd.envs import synthetic
  env = synthetic.SyntheticBatchEnv(n_env, episode_len=episode_len, ring=4)
  ids = torch.arange(n_env, dtype=torch.int32, device='cuda')
  count = torch.zeros(n_env, dtype=torch.int32, device='cuda')
  reset = torch.ones(n_env, dtype=torch.bool, device='cuda')
  action = torch.zeros(n_env, dtype=torch.int32, device='cuda')
  workers = list(range(n_env))
  for tick in range(steps):
    obs = env.step({'reset': reset})
    # episode step of every env, as the generator counts it (0 on a restart)
    count = torch.where(obs['is_first'], torch.zeros_like(count), count + 1)
    reset = obs['is_last']
    step = {**obs, 'action': action, 'env': ids, 'count': count,
            'tick': torch.full((n_env,), tick, dtype=torch.int32, device='cuda')}
    if extra:
      step.update(extra)
    rep.add_batch(step, workers)
  return steps


def _check_batch(batch, L, total_ticks, capacity, n_env, fifo=True):
  img = batch['image'].flatten(2)                                   # (B, L, 28224)
  env, count, tick = batch['env'].long(), batch['count'].long(), batch['tick'].long()
  salt = (env * 131 + count * 7)[..., None]
  want = ((salt + torch.arange(img.shape[-1], device=img.device)) & 0xFF).to(torch.uint8)
  assert torch.equal(img, want)                                     # every byte of every frame
  assert (env == env[:, :1]).all()                                  # one env per window
  assert (tick[:, 1:] == tick[:, :-1] + 1).all()                    # consecutive steps
  cont = count[:, 1:] == count[:, :-1] + 1
  restart = count[:, 1:] == 0
  assert (cont | restart).all()
  assert batch['is_first'][:, 0].all()                              # replay.py:279
  assert (batch['is_first'][:, 1:] == restart).all()
  # an episode end inside the window is flagged is_last on the step before a restart (replay.py:280-292)
  assert batch['is_last'][:, :-1][restart].all()
  # FIFO eviction (replay.py:171-191): nothing older than the newest `capacity` items
  if fifo:
    oldest_allowed = total_ticks - (capacity + n_env - 1) // n_env - L
    assert int(tick.min()) >= oldest_allowed
  sid = batch['stepid'].cpu().numpy().reshape(-1, L, 20)
  idx = sid[..., 16:].astype(np.int64)
  idx = (idx[..., 0] << 24) | (idx[..., 1] << 16) | (idx[..., 2] << 8) | idx[..., 3]
  step_on = (idx[:, 1:] == idx[:, :-1] + 1) | (idx[:, 1:] == 0)       # +1 inside a chunk, 0 in the next
  assert step_on.all()


def test_ppo_config_full_size_online(emb):
  """configs[1]: 64 envs, 84x84x4 u8, Replay(size 1e5, online, chunksize 1024),
  B=16, L=65."""
  n_env, L, B, cap = 64, 65, 16, 100_000
  rep = emb.Replay(length=L, capacity=cap, chunksize=1024, online=True, seed=0)
  total = _fill(emb, rep, n_env, (cap + L) // n_env + 3 * L)
  assert len(rep) == cap
  # Train mode serves the online queue first, oldest entry first, for as long as
  # the entry's first chunk exists -- also after its item was evicted
  # (replay.py:151-169), so the FIFO bound does not apply to these windows.
  first = rep.sample(B)
  _check_batch(first, L, total, cap, n_env, fifo=False)
  assert int(first['tick'].min()) == 1                               # windows start at step 1 (replay.py:114-118)
  for _ in range(8):
    _check_batch(rep.sample(B), L, total, cap, n_env, fifo=False)
  uniform = rep.sample(B, 'report')                                  # never the queue
  _check_batch(uniform, L, total, cap, n_env)
  stats = rep.stats()
  assert stats['items'] == cap and stats['inserts'] == total * n_env - n_env * (L - 1)


def test_ppo_shapes_through_the_per_step_host_path(emb):
  """The unchanged caller at BASELINE's shapes: one `add(step, worker)` per host
  step dict (64 workers, 84x84x4 uint8 frames from the host generator, length
  65) -- the C staging call, both pinned stage sets, many flushes and FIFO
  eviction -- then `sample(16)`: every gathered byte is the generator's."""
  from embodied_amd.envs import synthetic
  n_env, L, capacity, ticks = 64, 65, 2000, 140
  rep = emb.Replay(length=L, capacity=capacity, chunksize=1024, seed=3)
  envs = [synthetic.HostSyntheticEnv(e, episode_len=50) for e in range(n_env)]
  reset = [True] * n_env
  count = [0] * n_env
  for tick in range(ticks):
    for e, env in enumerate(envs):
      obs = env.step({'reset': reset[e]})
      count[e] = 0 if obs['is_first'] else count[e] + 1
      reset[e] = bool(obs['is_last'])
      rep.add({
          'image': obs['image'], 'reward': obs['reward'], 'is_first': np.bool_(obs['is_first']),
          'is_last': np.bool_(obs['is_last']), 'is_terminal': np.bool_(obs['is_terminal']),
          'action': np.int32(tick % 6), 'env': np.int32(e), 'count': np.int32(count[e]),
          'tick': np.int32(tick)}, e)
  assert rep._stage_plan is not None or emb.core.replay._add_step is None
  assert len(rep) == capacity
  for _ in range(6):
    _check_batch(rep.sample(16), L, ticks, capacity, n_env)


def test_dreamer_config_full_size_uniform_with_write_back(emb):
  """configs[2]: 10^6-step uniform replay, L=65, 40 KB of latents per step
  written back over sampled windows (dreamerv3/agent.py:144-150)."""
  n_env, L, T, B, cap = 64, 65, 64, 16, 1_000_000
  rep = emb.Replay(length=L, capacity=cap, chunksize=1024, seed=0)
  extra = {'dyn/deter': torch.zeros((n_env, 8192), device='cuda'),
           'dyn/stoch': torch.zeros((n_env, 32, 64), device='cuda')}
  total = _fill(emb, rep, n_env, (cap + L) // n_env + 2 * L, extra)
  assert len(rep) == cap
  for _ in range(4):
    _check_batch(rep.sample(B), L, total, cap, n_env)
  rows, _ = rep.sample_index(B)
  before = rep.gather(rows)
  assert not before['dyn/deter'].any()
  gen = torch.Generator(device='cuda').manual_seed(1)
  deter = torch.randn((B, T, 8192), device='cuda', generator=gen)
  stoch = torch.randn((B, T, 32, 64), device='cuda', generator=gen)
  rep.update({'stepid': before['stepid'][:, :T], 'dyn/deter': deter, 'dyn/stoch': stoch})
  after = rep.gather(rows)
  flat = rows[:, :T].reshape(-1)
  assert len(np.unique(flat)) == flat.size            # 16 windows in 10^6 steps: no overlap (seed 0)
  assert torch.equal(after['dyn/deter'][:, :T], deter)
  assert torch.equal(after['dyn/stoch'][:, :T], stoch)
  assert not after['dyn/deter'][:, T:].any()                          # the 65th step is not written
  assert torch.equal(after['image'], before['image'])


def test_crafter_config_256_envs_over_8_owners(emb):
  """configs[3] shape: 256 envs in 8 blocks of 32, 64x64x3 frames, L=65.  Eight
  'ranks' run in one process (the all-reduce of the packed batches is their
  sum): the merged batch equals what ONE replay over all 256 envs returns, each
  rank touching only its own block's rows."""
  from embodied_amd import distributed as D
  world, per, L, B, cap = 8, 32, 65, 16, 20_000
  kw = dict(chunksize=1024, seed=7)
  flats = {}
  shards = [
      D.ShardedReplay(L, cap, per, rank=r, world=world,
                      reduce=lambda flat, r=r: flats.__setitem__(r, flat.clone()), **kw)
      for r in range(world)]
  single = emb.Replay(L, cap, **kw)
  n = world * per
  ids = torch.arange(n, dtype=torch.int32, device='cuda')
  gen = torch.Generator(device='cuda').manual_seed(0)
  for tick in range(cap // n + 2 * L):
    step = {
        'image': torch.randint(0, 255, (n, 64, 64, 3), dtype=torch.uint8, device='cuda', generator=gen),
        'reward': torch.randn(n, device='cuda', generator=gen),
        'is_first': (ids + tick) % 97 == 0,
        'is_last': (ids + tick) % 97 == 96,
        'env': ids,
        'tick': torch.full((n,), tick, dtype=torch.int32, device='cuda'),
    }
    single.add_batch(step, list(range(n)))
    for r, shard in enumerate(shards):
      shard.add_batch({k: v[r * per:(r + 1) * per] for k, v in step.items()})
  assert len(single) == cap and all(len(s) == cap for s in shards)
  for _ in range(3):
    want = single.sample(B)
    for shard in shards:
      shard.sample(B)
    merged = sum(flats[r] for r in range(world))
    layout = D.PackedLayout(
        [(k.name, k.dtype, k.shape) for k in shards[0].replay._keys], B, L)
    got = layout.views(merged)
    for key in want:
      assert torch.equal(got[key], want[key]), key
    owners = (want['env'][:, 0] // per).tolist()
    for r in range(world):                       # a rank fills exactly the sequences it owns
      mine = layout.views(flats[r])['image'].flatten(1).any(1).tolist()
      assert mine == [o == r for o in owners]
