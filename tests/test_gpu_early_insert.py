"""The early insert (Replay.offer -> ops.obs_stack -> emb_replay_obs_stack_insert
-> emb_replay_publish): observation keys go to their pool rows in the launch
that builds the policy batch, the action follows after the policy.  Everything
the Replay hands out afterwards must be what the reference's Driver -> Replay.add
produces (embodied/core/driver.py:55-82, replay.py:77-118): checked against the
numpy oracle driven by the same envs and policy, and against the plain insert
path of this library.  Need a GPU."""
import numpy as np
import pytest
import torch

from oracle import np_oracle
from tests.conftest import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def emb():
  import embodied_amd
  assert torch.cuda.is_available(), 'these tests need the MI355X'
  return embodied_amd


def _host(batch):
  return {k: v.cpu().numpy() for k, v in batch.items()}


def _run_pair(emb, n, shape, length, capacity, chunksize, steps, online, stack, out_dtype=torch.bfloat16,
              layout='channels_first', extra_out=False, sample_every=7, episode_len=5, on_replay=None,
              block=1, unmasked=False, act_dtype=np.int32, ring=4):
  """Device Driver + Replay next to the oracle Driver + Replay on the same
  envs, policy and seeds; `stack` = the policy builds its batch with
  ops.obs_stack (which takes up the Driver's offer).  Returns the replay, the
  policy batches the agent saw and the number of early inserts."""
  from embodied_amd.envs import synthetic
  env = synthetic.SyntheticBatchEnv(n, shape=shape, episode_len=episode_len, ring=ring,
                                    takes_unmasked_actions=unmasked)
  rep = emb.Replay(length=length, capacity=capacity, chunksize=chunksize, online=online, seed=0)
  ref = np_oracle.Replay(length, capacity, chunksize, online, seed=0)
  if on_replay:
    on_replay(rep)
  hosts = [synthetic.HostSyntheticEnv(e, shape=shape, episode_len=episode_len) for e in range(n)]
  oracle = np_oracle.Driver(hosts)
  oracle.on_step(ref.add)
  driver = emb.Driver(batch_env=env, device='cuda')
  driver.on_step(rep.add)
  tick = {'dev': 0, 'host': 0}
  seen = []

  def acts_at(t):
    # negative values too: the mask is a multiply (-x -> -0 for floats)
    return ((np.arange(n) * 3 + t * 5) % 7 - 3).astype(act_dtype)

  def outs_at(t):
    return (np.arange(n * 6).reshape(n, 6) + t).astype(np.float32)

  def policy(carry, obs, **kw):
    t = tick['dev']
    tick['dev'] += 1
    if stack:
      batch = emb.ops.obs_stack(obs['image'], layout=layout, dtype=out_dtype, scale=1 / 255)
      seen.append((batch, obs['image'].clone()))
    outs = {'feat': torch.as_tensor(outs_at(t)).cuda()} if extra_out else {}
    return carry, {'action': torch.as_tensor(acts_at(t)).cuda()}, outs

  def host_policy(carry, obs):
    t = tick['host']
    tick['host'] += 1
    outs = {'feat': outs_at(t)} if extra_out else {}
    return carry, {'action': acts_at(t)}, outs

  driver.reset()
  # `block` steps per Driver call, nothing else touching the replay in between
  # (the early inserts inside a block take their rows from the publish before them)
  for t in range(0, steps, block):
    driver(policy, steps=n * block)
    for _ in range(block):
      oracle.step(host_policy)
    assert len(rep) == len(ref), t
    if len(ref) and (t // block) % sample_every == 0:
      mode = 'train' if (t // block) % 2 else 'report'
      assert_same(_host(rep.sample(3, mode)), ref.sample(3, mode), f'step {t}')
  return rep, ref, seen


@pytest.mark.parametrize('online', [False, True])
@pytest.mark.parametrize('shape', [(8, 8, 4), (8, 8, 3), (4, 8, 2), (4, 4, 1)])
def test_early_insert_matches_oracle(emb, online, shape):
  """Small frames, chunks of 8 rows (many rotations), capacity churn, the online
  queue: every sampled batch equals the oracle's, with the early insert taken on
  every step after the first."""
  n, steps = 5, 90
  rep, ref, seen = _run_pair(emb, n, shape, length=4, capacity=60, chunksize=8, steps=steps,
                             online=online, stack=True)
  assert rep.early_inserts == steps - 1       # the first step opens the workers' chunks
  # ... and their publishes left the bookkeeping to the helper thread (conftest
  # lifts the pace condition; EMB_DEFER_INDEX=0 runs in a child process)
  import os
  if os.environ.get('EMB_DEFER_INDEX') != '0':
    assert rep.profile_report('deferred')[0] >= steps - 3      # (the first one has no pace yet)
  got, want = rep.stats(), ref.stats()
  for k in ('items', 'chunks', 'streams', 'inserts', 'samples'):
    assert got[k] == want[k], k
  # the policy batches are what the plain obs stack produces
  for batch, frames in seen[::9]:
    want = emb.ops.obs_stack(frames, layout='channels_first', dtype=torch.bfloat16, scale=1 / 255)
    assert torch.equal(batch.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize('out_dtype,layout', [
    (torch.uint8, 'same'), (torch.float32, 'channels_first'), (torch.float16, 'same')])
def test_early_insert_other_policy_layouts(emb, out_dtype, layout):
  rep, ref, seen = _run_pair(emb, 6, (8, 12, 4), length=5, capacity=80, chunksize=16, steps=60,
                             online=True, stack=True, out_dtype=out_dtype, layout=layout)
  assert rep.early_inserts == 59
  for batch, frames in seen[::7]:
    want = emb.ops.obs_stack(frames, layout=layout, dtype=out_dtype, scale=1 / 255)
    assert torch.equal(batch, want)


@pytest.mark.parametrize('shape', [(200, 200, 3), (256, 160, 4), (84, 84, 4)])
def test_early_insert_frames_of_many_blocks(emb, shape):
  """Frames wider than the launch's 32 frame blocks per env (the blocks loop),
  and BASELINE's 84 x 84 x 4 (7 blocks + the narrow workgroup per env on the
  one-dimensional grid): every sampled batch equals the oracle's, the policy
  batch equals the plain obs stack."""
  rep, ref, seen = _run_pair(emb, 3, shape, length=3, capacity=40, chunksize=8, steps=30,
                             online=True, stack=True, out_dtype=torch.bfloat16, layout='channels_first')
  assert rep.early_inserts == 29
  for batch, frames in seen[::5]:
    want = emb.ops.obs_stack(frames, layout='channels_first', dtype=torch.bfloat16, scale=1 / 255)
    assert torch.equal(batch, want)


def test_early_insert_with_agent_outputs(emb):
  """A policy that also returns replay outputs (wider keys: the general insert
  launch carries action + outputs, the observation keys are already in place)."""
  rep, ref, _ = _run_pair(emb, 4, (8, 8, 4), length=3, capacity=50, chunksize=8, steps=70,
                          online=False, stack=True, extra_out=True)
  assert rep.early_inserts == 69


def test_full_size_early_insert_equals_plain_insert(emb):
  """BASELINE shapes (64 envs, 84x84x4, L=65, chunksize 1024): two replays fed by
  the same envs, one through the early insert, one through the plain fused
  insert; identical pools as seen through identical samples, and equal to the
  oracle's flags / step ids."""
  n, L = 64, 65
  a, ref, _ = _run_pair(emb, n, (84, 84, 4), length=L, capacity=3000, chunksize=1024, steps=150,
                        online=True, stack=True, sample_every=50, episode_len=40)
  b, _, _ = _run_pair(emb, n, (84, 84, 4), length=L, capacity=3000, chunksize=1024, steps=150,
                      online=True, stack=False, sample_every=50, episode_len=40)
  assert a.early_inserts == 149 and b.early_inserts == 0
  for _ in range(3):
    assert_same(_host(a.sample(16)), _host(b.sample(16)), 'early vs plain')


@pytest.mark.parametrize('n', [150, 200])
def test_many_envs_tables(emb, n):
  """The per-env table (rows, step ids, narrow keys) reaches the launch in device
  memory: written through the BAR while it fits a 4 KiB slot of the argument ring
  (<= 162 envs), through the pinned staging ring + one copy beyond that."""
  rep, ref, _ = _run_pair(emb, n, (4, 4, 4), length=3, capacity=2000, chunksize=8, steps=40,
                          online=False, stack=True, sample_every=5)
  assert rep.early_inserts == 39


def test_checkpoint_between_early_insert_and_publish(emb, tmp_path):
  """save() inside the policy closes every open chunk after the observation keys
  went to their (old) rows: the publish notices that its rows are not the
  reserved ones and writes the whole step; nothing of the stale rows shows."""
  from embodied_amd.envs import synthetic
  n, shape = 4, (8, 8, 4)
  env = synthetic.SyntheticBatchEnv(n, shape=shape, episode_len=6, ring=4)
  rep = emb.Replay(length=3, capacity=60, chunksize=8, seed=0, directory=tmp_path, save_wait=True)
  ref = np_oracle.Replay(3, 60, 8, False, seed=0)
  hosts = [synthetic.HostSyntheticEnv(e, shape=shape, episode_len=6) for e in range(n)]
  oracle = np_oracle.Driver(hosts)
  oracle.on_step(ref.add)
  driver = emb.Driver(batch_env=env, device='cuda')
  driver.on_step(rep.add)
  tick = [0, 0]

  def policy(carry, obs, **kw):
    emb.ops.obs_stack(obs['image'], dtype=torch.float32)
    tick[0] += 1
    if tick[0] % 7 == 3:
      rep.save()
    return carry, {'action': torch.full((n,), tick[0], dtype=torch.int32, device='cuda')}, {}

  def host_policy(carry, obs):
    tick[1] += 1
    return carry, {'action': np.full(n, tick[1], np.int32)}, {}

  for t in range(40):
    driver(policy, steps=n)
    oracle.step(host_policy)
    if len(ref) and t % 3 == 0:
      got, want = _host(rep.sample(4)), ref.sample(4)
      # (closing chunks early changes chunk ids and rows, not contents: compare payload)
      for key in ('image', 'reward', 'action', 'is_first', 'is_last'):
        assert np.array_equal(got[key], want[key]), (t, key)


def test_offer_not_taken_or_stale_is_harmless(emb):
  """An agent that stacks only now and then, stacks twice, or stacks a tensor of
  an older step: the replay still equals the oracle."""
  from embodied_amd.envs import synthetic
  n, shape = 4, (8, 8, 4)
  env = synthetic.SyntheticBatchEnv(n, shape=shape, episode_len=6, ring=2)
  rep = emb.Replay(length=3, capacity=40, chunksize=8, seed=0)
  ref = np_oracle.Replay(3, 40, 8, False, seed=0)
  hosts = [synthetic.HostSyntheticEnv(e, shape=shape, episode_len=6) for e in range(n)]
  oracle = np_oracle.Driver(hosts)
  oracle.on_step(ref.add)
  driver = emb.Driver(batch_env=env, device='cuda')
  driver.on_step(rep.add)
  tick = [0]
  old = []

  def policy(carry, obs, **kw):
    t = tick[0]
    tick[0] += 1
    if t % 3 == 0:
      emb.ops.obs_stack(obs['image'], dtype=torch.float32)
    if t % 6 == 0:
      emb.ops.obs_stack(obs['image'], dtype=torch.float32)      # second call: plain stack
    if t % 5 == 1 and old:
      emb.ops.obs_stack(old[-1], dtype=torch.float32)           # another step's tensor (ring of 2)
    old.append(obs['image'])
    return carry, {'action': torch.full((n,), t, dtype=torch.int32, device='cuda')}, {}

  host_tick = [0]

  def host_policy(carry, obs):
    t = host_tick[0]
    host_tick[0] += 1
    return carry, {'action': np.full(n, t, np.int32)}, {}

  driver.reset()
  for t in range(60):
    driver(policy, steps=n)
    oracle.step(host_policy)
    if len(ref) and t % 4 == 0:
      assert_same(_host(rep.sample(5)), ref.sample(5), f'step {t}')


def test_publish_survives_pool_growth(emb):
  """The pool fills up between the early insert and the publish: the publish
  raises PoolFull inside, the pool grows (device-to-device copy behind the early
  insert on the same stream) and the retry keeps the early rows."""
  rep, ref, _ = _run_pair(emb, 6, (8, 8, 4), length=4, capacity=None, chunksize=4, steps=120,
                          online=False, stack=True, sample_every=6)
  assert rep.early_inserts == 119


@pytest.mark.parametrize('parallel', [False, True])
def test_early_insert_with_host_envs(emb, parallel):
  """Real simulators on the host (in-process, and one process per env writing
  into the shared observation slab): the uploaded step is offered to the replay
  as well, the agent's obs stack on the uploaded frames is the early insert."""
  from embodied_amd.envs import synthetic
  n, shape = 4, (8, 8, 4)
  fns = [(lambda e=e: synthetic.HostSyntheticEnv(e, shape=shape, episode_len=6)) for e in range(n)]
  driver = emb.Driver(fns, parallel=parallel, device='cuda')
  rep = emb.Replay(length=3, capacity=50, chunksize=8, online=True, seed=0)
  ref = np_oracle.Replay(3, 50, 8, True, seed=0)
  oracle = np_oracle.Driver([synthetic.HostSyntheticEnv(e, shape=shape, episode_len=6) for e in range(n)])
  oracle.on_step(ref.add)
  driver.on_step(rep.add)
  tick = [0, 0]

  def policy(carry, obs, **kw):
    batch = emb.ops.obs_stack(obs['image'], layout='channels_first', dtype=torch.float32, scale=1 / 255)
    assert torch.equal(batch, obs['image'].permute(0, 3, 1, 2).float() / 255)
    tick[0] += 1
    return carry, {'action': np.full(n, tick[0], np.int32)}, {}

  def host_policy(carry, obs):
    tick[1] += 1
    return carry, {'action': np.full(n, tick[1], np.int32)}, {}

  try:
    driver.reset()
    for t in range(45):
      driver(policy, steps=n)
      oracle.step(host_policy)
      if len(ref) and t % 4 == 0:
        mode = 'train' if t % 8 else 'report'
        assert_same(_host(rep.sample(3, mode)), ref.sample(3, mode), f'step {t}')
    assert rep.early_inserts == 44
  finally:
    driver.close()


def test_deferred_bookkeeping_with_readers_on_other_threads(emb):
  """emb_replay_publish hands its index bookkeeping to the library's helper
  thread (abi.cpp DeferGate); every later operation on the replay OR on its
  selector handle waits for it first.  Two threads read the replay's and the
  selector's length all the time while the loop runs: every length they see is
  one the oracle passes through, and every sampled batch equals the oracle's."""
  import threading
  box, seen_lengths, stop = [], set(), threading.Event()

  import time

  def reader(via_selector):
    while not stop.is_set():
      if box:
        seen_lengths.add(len(box[0].sampler) if via_selector else len(box[0]))
      time.sleep(0)       # (a full replay answers len() without a library call: let go of the GIL)

  threads = [threading.Thread(target=reader, args=(flag,), daemon=True) for flag in (False, True)]
  [t.start() for t in threads]
  try:
    rep, ref, _ = _run_pair(emb, 16, (8, 8, 4), length=4, capacity=300, chunksize=8, steps=400,
                            online=True, stack=True, sample_every=3, on_replay=box.append)
  finally:
    stop.set()
    [t.join() for t in threads]
  assert rep.early_inserts == 399
  import os
  if os.environ.get('EMB_DEFER_INDEX') != '0':
    # (the self-check may pause deferral when readers make every job a wait)
    assert rep.profile_report('deferred')[0] > 50
  assert len(seen_lengths) > 3 and max(seen_lengths) <= 300
  # lengths grow by whole steps of the 16 workers until the capacity is reached
  assert all(x % 16 == 0 or x == 300 for x in seen_lengths), sorted(seen_lengths)[:20]


@pytest.mark.parametrize('seed', range(8))
def test_early_insert_random_configurations_against_oracle(emb, seed):
  """Random replay geometry (length, capacity, chunk size, env count, episode
  length, online queue, sampling cadence) through the early insert and the
  deferred bookkeeping: every sampled batch, every length and the final
  statistics equal the oracle's."""
  rng = np.random.default_rng(1000 + seed)
  n = int(rng.integers(2, 24))
  length = int(rng.integers(1, 7))
  chunksize = int(rng.integers(max(2, length // 2), 20))
  capacity = int(rng.integers(max(n, 8), 400))
  steps = int(rng.integers(40, 160))
  shape = [(8, 8, 4), (4, 8, 2), (8, 4, 1), (4, 4, 4)][int(rng.integers(0, 4))]
  rep, ref, _ = _run_pair(
      emb, n, shape, length=length, capacity=capacity, chunksize=chunksize, steps=steps,
      online=bool(rng.integers(0, 2)), stack=True, sample_every=int(rng.integers(1, 9)),
      episode_len=int(rng.integers(2, 12)), extra_out=bool(rng.integers(0, 2)))
  assert rep.early_inserts == steps - 1
  got, want = rep.stats(), ref.stats()
  for k in ('items', 'chunks', 'streams', 'inserts', 'samples'):
    assert got[k] == want[k], (k, got[k], want[k])


def test_env_processes_forked_while_the_helper_thread_is_running(emb):
  """A parallel host-env Driver is created (its env processes are forked) after
  a device loop has started the library's helper thread: fork waits for the job
  in flight, the children never wait for a helper they do not have, and both
  loops go on with oracle parity."""
  from embodied_amd.envs import synthetic
  rep, ref, _ = _run_pair(emb, 8, (8, 8, 4), length=3, capacity=100, chunksize=8, steps=40,
                          online=False, stack=True)
  n, shape = 4, (8, 8, 4)
  fns = [(lambda e=e: synthetic.HostSyntheticEnv(e, shape=shape, episode_len=6)) for e in range(n)]
  driver = emb.Driver(fns, parallel=True, device='cuda')          # forks here
  rep2 = emb.Replay(length=3, capacity=50, chunksize=8, seed=1)
  ref2 = np_oracle.Replay(3, 50, 8, False, seed=1)
  oracle = np_oracle.Driver([synthetic.HostSyntheticEnv(e, shape=shape, episode_len=6) for e in range(n)])
  oracle.on_step(ref2.add)
  driver.on_step(rep2.add)
  tick = [0, 0]

  def policy(carry, obs, **kw):
    emb.ops.obs_stack(obs['image'], dtype=torch.float32)
    tick[0] += 1
    return carry, {'action': np.full(n, tick[0], np.int32)}, {}

  def host_policy(carry, obs):
    tick[1] += 1
    return carry, {'action': np.full(n, tick[1], np.int32)}, {}

  try:
    driver.reset()
    for t in range(20):
      driver(policy, steps=n)
      oracle.step(host_policy)
    assert_same(_host(rep2.sample(4)), ref2.sample(4), 'forked driver')
    assert_same(_host(rep.sample(4)), ref.sample(4), 'first replay after the fork')
  finally:
    driver.close()


@pytest.mark.parametrize('flags', [(), ('--unmasked',)])
def test_soak_stepping_loop_with_sampler_threads(flags):
  """tools/soak_early_insert.py for a few seconds: the device Driver steps
  through the early insert and the helper thread while two sampler threads draw
  on their own streams; every sampled window must follow the env's generator.
  `--unmasked`: with the action's pool write carried into the next early insert
  (settled by whichever thread's sample needs it first)."""
  import pathlib
  import subprocess
  import sys
  root = pathlib.Path(__file__).resolve().parent.parent
  res = subprocess.run([sys.executable, str(root / 'tools' / 'soak_early_insert.py'), '--seconds', '4', *flags],
                       cwd=root, capture_output=True, text=True, timeout=300)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
  assert 'errors []' in res.stdout, res.stdout[-2000:]


@pytest.mark.parametrize('seed', range(6))
def test_blocks_of_steps_per_driver_call(emb, seed):
  """Several steps per Driver call with nothing else touching the replay in
  between (publish -> next early insert back to back, the helper thread at the
  bookkeeping meanwhile): small chunks, so that rotations fall inside blocks;
  samples after every block equal the oracle's."""
  import os
  rng = np.random.default_rng(2000 + seed)
  n = int(rng.integers(2, 20))
  length = int(rng.integers(1, 6))
  chunksize = int(rng.integers(max(3, length), 24))
  block = int(rng.integers(3, 12))
  steps = block * int(rng.integers(8, 20))
  rep, ref, _ = _run_pair(
      emb, n, (8, 8, 4), length=length, capacity=int(rng.integers(max(n, 8), 300)), chunksize=chunksize,
      steps=steps, online=bool(rng.integers(0, 2)), stack=True, sample_every=int(rng.integers(1, 4)),
      episode_len=int(rng.integers(2, 12)), block=block)
  assert rep.early_inserts == steps - 1
  if os.environ.get('EMB_DEFER_INDEX') != '0':
    assert rep.profile_report('deferred')[0] >= steps - 3
  got, want = rep.stats(), ref.stats()
  for k in ('items', 'chunks', 'streams', 'inserts', 'samples'):
    assert got[k] == want[k], (k, got[k], want[k])


# ------------------------------------------------------------ carried publish --

@pytest.mark.parametrize('online', [False, True])
@pytest.mark.parametrize('sample_every', [1, 7])
def test_carried_publish_matches_oracle(emb, online, sample_every):
  """An env that takes unmasked actions (Driver: no masked copy, Replay.
  carry_publish): the action's pool write rides in the NEXT step's early-insert
  launch, or is settled by a launch of its own when a sample comes first.  Every
  sampled batch equals the oracle's (which masks the stored action,
  driver.py:72-74); with a sample every 7th step most publishes ride along."""
  n, steps = 5, 90
  rep, ref, _ = _run_pair(emb, n, (8, 8, 4), length=4, capacity=60, chunksize=8, steps=steps,
                          online=online, stack=True, sample_every=sample_every, unmasked=True)
  assert rep.early_inserts == steps - 1
  inline, total = rep.profile_report('carried')[:2]
  assert total >= steps - 2                       # every publish behind an early insert was carried
  # (a sample settles a carried publish only if one of its windows ends on the
  # newest step of a worker -- the online queue's fresh windows do, most uniform
  # draws do not)
  assert inline >= total * (0.7 if sample_every == 7 else 0.2)
  if online and sample_every == 1:
    assert inline < total                         # some train-mode samples took the freshest windows
  got, want = rep.stats(), ref.stats()
  for k in ('items', 'chunks', 'streams', 'inserts', 'samples'):
    assert got[k] == want[k], k
  assert_same(_host(rep.sample(6)), ref.sample(6), 'final')


@pytest.mark.parametrize('ring', [1, 2, 0])
def test_carried_publish_with_an_env_that_rewrites_its_flags_in_place(emb, ring):
  """ring=1: the env writes every step's observations -- is_last among them --
  into the same buffers, so by the time the carried action is written (in the
  NEXT step's early-insert launch, behind the next env step) the env's flag
  buffer holds the next step's flags.  The mask reads the step's own is_last
  from the replay's rows, where that step's early insert stored it: stored
  actions equal the oracle's whatever the env does with its buffers (ring=0:
  fresh tensors each step).  Env e's episodes last 3 + 13 * (e % 8) steps: ends
  fall on different steps for different envs, so flags of the wrong step would
  mask the wrong actions."""
  n, steps = 6, 80
  rep, ref, _ = _run_pair(emb, n, (8, 8, 4), length=3, capacity=300, chunksize=8, steps=steps,
                          online=False, stack=True, sample_every=9, unmasked=True, episode_len=3, ring=ring)
  inline, total = rep.profile_report('carried')[:2]
  assert total >= steps - 2 and inline >= total * 0.7       # the publishes WERE carried
  for _ in range(5):
    got, want = _host(rep.sample(16)), ref.sample(16)
    assert np.array_equal(got['action'], want['action'])
    assert (want['action'][want['is_last']] == 0).all() and (want['action'] != 0).any()
    assert_same(got, want, f'ring {ring}')


@pytest.mark.parametrize('act_dtype', [np.float32, np.float16, np.int64, np.uint8])
def test_carried_publish_masks_in_the_actions_own_dtype(emb, act_dtype):
  """value * ~is_last in the key's dtype inside the carrying launch: -x -> -0.0
  for floats, exactly as the publish launch and numpy do."""
  rep, ref, _ = _run_pair(emb, 6, (8, 8, 4), length=3, capacity=200, chunksize=16, steps=60,
                          online=False, stack=True, sample_every=11, unmasked=True, act_dtype=act_dtype)
  assert rep.profile_report('carried')[0] > 40
  for _ in range(4):
    got, want = _host(rep.sample(9)), ref.sample(9)
    assert got['action'].dtype == want['action'].dtype
    assert np.array_equal(got['action'].view(np.uint8), want['action'].view(np.uint8))     # bit for bit (-0.0)
    assert_same(got, want, 'dtype')


def test_carried_publish_with_pool_growth_blocks_and_agent_outputs(emb):
  """Pool growth between a carried publish and the launch that would have taken
  it along (settled before the pool is copied), several steps per Driver call,
  and agent outputs beside the action (more than one key left: nothing to
  carry, same results)."""
  rep, ref, _ = _run_pair(emb, 6, (8, 8, 4), length=4, capacity=None, chunksize=4, steps=120,
                          online=False, stack=True, sample_every=9, unmasked=True)
  assert_same(_host(rep.sample(8)), ref.sample(8), 'growth')
  rep, ref, _ = _run_pair(emb, 4, (8, 8, 4), length=3, capacity=80, chunksize=8, steps=60, online=True,
                          stack=True, sample_every=3, unmasked=True, block=3)
  assert_same(_host(rep.sample(8)), ref.sample(8), 'blocks')
  rep, ref, _ = _run_pair(emb, 4, (8, 8, 4), length=3, capacity=80, chunksize=8, steps=60, online=False,
                          stack=True, sample_every=5, unmasked=True, extra_out=True)
  assert rep.profile_report('carried')[1] == 0
  assert_same(_host(rep.sample(8)), ref.sample(8), 'outs')


def test_carried_publish_is_settled_before_checkpoints_and_on_request(emb, tmp_path):
  """save() reads the pool through torch: the chunk bookkeeping call in front of
  it settles a carried publish; carry_publish(False) settles too."""
  from embodied_amd.envs import synthetic
  n = 4
  env = synthetic.SyntheticBatchEnv(n, shape=(8, 8, 4), episode_len=50, ring=4, takes_unmasked_actions=True)
  rep = emb.Replay(length=2, capacity=400, chunksize=16, seed=0, directory=str(tmp_path), save_wait=True)
  driver = emb.Driver(batch_env=env, device='cuda')
  driver.on_step(rep.add)
  tick = [0]

  def policy(carry, obs, **kw):
    emb.ops.obs_stack(obs['image'], layout='channels_first', dtype=torch.bfloat16, scale=1 / 255)
    tick[0] += 1
    return carry, {'action': torch.full((n,), tick[0], dtype=torch.int32, device='cuda')}, {}

  driver.reset()
  driver(policy, steps=n)                         # the first step allocates the pool (uninitialised memory)
  key = rep._keys[rep._keyid['action']]
  torch.cuda.synchronize()
  key.pool.zero_()
  driver(policy, steps=n * 9)
  assert rep.profile_report('carried', reset=False)[1] >= 8
  rep.save()
  torch.cuda.synchronize()
  stored = key.pool.view(torch.int32).cpu().numpy()
  assert sorted(set(stored[stored > 0].tolist())) == list(range(2, 11))      # the 10th step's action is in
  driver(policy, steps=n * 3)
  rep.carry_publish(False)
  torch.cuda.synchronize()
  stored = key.pool.view(torch.int32).cpu().numpy()
  assert stored.max() == 13
  again = emb.Replay(length=2, capacity=400, chunksize=16, seed=0, directory=str(tmp_path))
  again.load()
  assert len(again) >= 4 * 8


def test_carried_publish_with_actions_that_need_a_contiguous_copy(emb):
  """A policy that returns a broadcast view (stride 0): the insert works on a
  contiguous copy, which has to outlive the call when the write is carried."""
  from embodied_amd.envs import synthetic
  n, shape = 6, (8, 8, 4)
  env = synthetic.SyntheticBatchEnv(n, shape=shape, episode_len=4, ring=4, takes_unmasked_actions=True)
  rep = emb.Replay(length=3, capacity=300, chunksize=16, seed=0)
  ref = np_oracle.Replay(3, 300, 16, False, seed=0)
  oracle = np_oracle.Driver([synthetic.HostSyntheticEnv(e, shape=shape, episode_len=4) for e in range(n)])
  oracle.on_step(ref.add)
  driver = emb.Driver(batch_env=env, device='cuda')
  driver.on_step(rep.add)
  ticks = torch.arange(1, 200, dtype=torch.int32, device='cuda')
  tick = [0, 0]

  def policy(carry, obs, **kw):
    emb.ops.obs_stack(obs['image'], layout='channels_first', dtype=torch.bfloat16, scale=1 / 255)
    tick[0] += 1
    junk = torch.full((n,), -1, dtype=torch.int32, device='cuda')      # allocator traffic around the copy
    del junk
    return carry, {'action': ticks[tick[0]].expand(n)}, {}

  def host_policy(carry, obs):
    tick[1] += 1
    return carry, {'action': np.full(n, tick[1] + 1, np.int32)}, {}

  driver.reset()
  for t in range(60):
    driver(policy, steps=n)
    torch.full((n,), -7, dtype=torch.int32, device='cuda')              # reuses freed blocks, if any
    oracle.step(host_policy)
    if t % 13 == 12:
      assert_same(_host(rep.sample(7)), ref.sample(7), f'step {t}')
  assert rep.profile_report('carried')[0] > 40
  assert_same(_host(rep.sample(12)), ref.sample(12), 'final')


def test_early_inserts_on_predicted_rows(emb):
  """A stepping loop that calls nothing else on the replay: the publish that
  hands its bookkeeping to the helper thread also says where the next step goes
  (every cursor one row on), and the next early insert does not wait for the
  helper.  Chunks of 16 rows: a prediction is made on 15 steps of 16.  The
  replay must equal the oracle's after 300 such steps, and after sampling in
  between (which invalidates a prediction)."""
  import os
  if os.environ.get('EMB_DEFER_INDEX') == '0':
    pytest.skip('no helper thread in this run')
  from embodied_amd.envs import synthetic
  n, shape = 6, (8, 8, 4)
  for unmasked in (False, True):
    env = synthetic.SyntheticBatchEnv(n, shape=shape, episode_len=7, ring=4, takes_unmasked_actions=unmasked)
    rep = emb.Replay(length=4, capacity=500, chunksize=16, online=True, seed=0)
    ref = np_oracle.Replay(4, 500, 16, True, seed=0)
    oracle = np_oracle.Driver([synthetic.HostSyntheticEnv(e, shape=shape, episode_len=7) for e in range(n)])
    oracle.on_step(ref.add)
    driver = emb.Driver(batch_env=env, device='cuda')
    driver.on_step(rep.add)
    tick = [0, 0]

    def policy(carry, obs, **kw):
      emb.ops.obs_stack(obs['image'], layout='channels_first', dtype=torch.bfloat16, scale=1 / 255)
      tick[0] += 1
      return carry, {'action': torch.full((n,), tick[0] % 5 - 2, dtype=torch.int32, device='cuda')}, {}

    def host_policy(carry, obs):
      tick[1] += 1
      return carry, {'action': np.full(n, tick[1] % 5 - 2, np.int32)}, {}

    driver.reset()
    driver(policy, steps=n * 300)
    for _ in range(300):
      oracle.step(host_policy)
    deferred, predicted = rep.profile_report('deferred')[:2]
    assert deferred >= 295 and predicted >= 270, (deferred, predicted)
    assert len(rep) == len(ref)
    for mode in ('train', 'report', 'train'):
      assert_same(_host(rep.sample(8, mode)), ref.sample(8, mode), f'after 300 steps, {mode}')
    for t in range(40):
      driver(policy, steps=n)
      oracle.step(host_policy)
      if t % 3 == 0:
        assert_same(_host(rep.sample(5)), ref.sample(5), f'step {t}')
