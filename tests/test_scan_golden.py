"""Return scans against fixtures produced by EXECUTING the reference's own
source (oracle/gen_scan_golden.py -> tests/golden/scan_*.npz).

CPU: the numpy oracle equals the fixtures exactly in float32 (same operations
in the same order).  GPU: the HIP scans are within 1e-5 (north_star's
float-return tolerance; the kernels use a parallel affine scan with FMA, which
rounds differently from the sequential loop at ~1e-7 relative).
"""
import numpy as np
import pytest

from oracle import np_oracle
from tests import scan_cases as cases
from tests.conftest import GOLDEN

TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope='module')
def batch_major():
  with np.load(GOLDEN / 'scan_batch_major.npz') as f:
    return {k: f[k] for k in f.files}


@pytest.fixture(scope='module')
def director():
  with np.load(GOLDEN / 'scan_director.npz') as f:
    return {k: f[k] for k in f.files}


def tag(seed, shape):
  return f's{seed}_{shape[0]}x{shape[1]}'


def lambda_sets(seed):
  return [(i, p) for i, p in enumerate(cases.LAMBDA_PARAMS) if not (i and seed > 1)]


BT = [(seed, shape) for seed in cases.SEEDS for shape in cases.SHAPES_BT]
TB = [(seed, shape) for seed in cases.SEEDS for shape in cases.SHAPES_TB]
TRAJ = [(seed, shape) for seed in cases.SEEDS[:2] for shape in cases.TRAJ_SHAPES_TB]


def test_fixture_inputs_are_the_seeded_inputs(batch_major, director):
  for seed, shape in BT:
    assert np.array_equal(batch_major['in_' + tag(seed, shape)],
                          cases.digest(cases.batch_major(seed, shape)))
  for seed, shape in TB:
    assert np.array_equal(director['in_' + tag(seed, shape)],
                          cases.digest(cases.time_major(seed, shape)))
  for seed, shape in TRAJ:
    assert np.array_equal(director['trajin_' + tag(seed, shape)],
                          cases.digest(cases.trajectory(seed, shape)))


@pytest.mark.parametrize('seed,shape', BT)
def test_oracle_gae_and_lambda_equal_reference_fixture(batch_major, seed, shape):
  inp = cases.batch_major(seed, shape)
  adv, tar = np_oracle.gae(inp['rew'], inp['val'], inp['last'], inp['term'], **cases.GAE_PARAMS)
  assert adv.dtype == np.float32
  assert np.array_equal(adv, batch_major['gae_adv_' + tag(seed, shape)])
  assert np.array_equal(tar, batch_major['gae_tar_' + tag(seed, shape)])
  for i, params in lambda_sets(seed):
    ret = np_oracle.lambda_return(inp['last'], inp['term'], inp['rew'], inp['boot'], **params)
    assert np.array_equal(ret, batch_major[f'lambda{i}_' + tag(seed, shape)])


@pytest.mark.parametrize('seed,shape', TB)
def test_oracle_director_score_equals_reference_fixture(director, seed, shape):
  inp = cases.time_major(seed, shape)
  ret = np_oracle.director_score(inp['rew'], inp['cont'], inp['value'], **cases.DIRECTOR_PARAMS)
  assert np.array_equal(ret, director['score_' + tag(seed, shape)])


@pytest.mark.parametrize('seed,shape', TRAJ)
def test_oracle_split_and_abstract_equal_reference_fixture(director, seed, shape):
  traj = cases.trajectory(seed, shape)
  k = cases.SKILL_DURATION
  for key, x in traj.items():
    reward = key.startswith('reward_')
    want = director[f'split_{key}_' + tag(seed, shape)]
    assert np.array_equal(np_oracle.split_traj(x, k, reward), want), key
    kind = 'reward' if reward else 'cont' if key == 'cont' else 'first'
    want = director[f'abstract_{key}_' + tag(seed, shape)]
    assert np.array_equal(np_oracle.abstract_traj(x, traj['cont'], k, kind), want), key


@pytest.mark.reference
def test_scan_fixtures_are_current(batch_major, director, tmp_path, monkeypatch):
  """Build container only: re-running the generator on /root/reference gives
  the committed fixtures."""
  from oracle import refload
  if not refload.available():
    pytest.skip('no /root/reference here')
  from oracle import gen_scan_golden as gen
  monkeypatch.setattr(gen, 'ROOT', tmp_path)
  (tmp_path / 'tests').mkdir()
  gen.main()
  for name, want in (('scan_batch_major', batch_major), ('scan_director', director)):
    with np.load(tmp_path / 'tests' / 'golden' / f'{name}.npz') as f:
      assert set(f.files) == set(want)
      for key in f.files:
        assert np.array_equal(f[key], want[key]), key


# ------------------------------------------------------------------ HIP path --

@pytest.fixture(scope='module')
def emb():
  import torch
  if not torch.cuda.is_available():
    pytest.skip('needs a GPU')
  import embodied_amd
  return embodied_amd


def dev(x):
  import torch
  return torch.as_tensor(np.ascontiguousarray(x)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize('seed,shape', BT)
def test_hip_gae_and_lambda_match_reference_fixture(emb, batch_major, seed, shape):
  inp = cases.batch_major(seed, shape)
  adv, tar = emb.scans.gae(dev(inp['rew']), dev(inp['val']), dev(inp['last']), dev(inp['term']),
                           **cases.GAE_PARAMS)
  np.testing.assert_allclose(adv.cpu().numpy(), batch_major['gae_adv_' + tag(seed, shape)], **TOL)
  np.testing.assert_allclose(tar.cpu().numpy(), batch_major['gae_tar_' + tag(seed, shape)], **TOL)
  for i, params in lambda_sets(seed):
    ret = emb.scans.lambda_return(dev(inp['last']), dev(inp['term']), dev(inp['rew']),
                                  dev(inp['val']), dev(inp['boot']), **params)
    np.testing.assert_allclose(ret.cpu().numpy(), batch_major[f'lambda{i}_' + tag(seed, shape)], **TOL)


@pytest.mark.gpu
def test_hip_lambda_returns_in_one_launch_match_reference_fixture(emb, batch_major):
  """scans.lambda_returns: the replay-shaped (16,64) and the imagination-shaped
  (1024,16) problems of a DreamerV3 train step in ONE launch, with different
  disc / lam per problem -- each result within 1e-5 of the reference fixture (and
  of the single-problem launch, which may pick another kernel form); 1 to 4
  problems per launch, long rows and more than four problems (one launch each),
  an empty problem, caller-owned outputs."""
  import torch
  problems, wants, singles = [], [], []
  for seed, shape in BT:
    inp = cases.batch_major(seed, shape)
    i, params = lambda_sets(seed)[-1]
    args = (dev(inp['last']), dev(inp['term']), dev(inp['rew']), dev(inp['val']), dev(inp['boot']))
    problems.append((*args, params['disc'], params['lam']))
    wants.append(batch_major[f'lambda{i}_' + tag(seed, shape)])
    singles.append(emb.scans.lambda_return(*args, **params))
  shapes = {tuple(p[2].shape) for p in problems}
  assert (16, 64) in shapes and (1024, 16) in shapes
  short = [i for i, p in enumerate(problems) if p[2].shape[1] <= 257]
  assert len(short) >= 4
  for pick in (short[:1], short[:2], short[:3], short[:4], short, list(range(len(problems)))):
    got = emb.scans.lambda_returns([problems[i] for i in pick])
    assert len(got) == len(pick)
    for ret, i in zip(got, pick):
      np.testing.assert_allclose(ret.cpu().numpy(), wants[i], **TOL)
      np.testing.assert_allclose(ret.cpu().numpy(), singles[i].cpu().numpy(), **TOL)
  own = [torch.empty_like(s) for s in singles[:2]]
  got = emb.scans.lambda_returns(problems[:2], out=own)
  assert got[0] is own[0]
  np.testing.assert_allclose(own[1].cpu().numpy(), wants[1], **TOL)
  with pytest.raises(ValueError):
    emb.scans.lambda_returns(problems[:2], out=[own[0], own[0][:, :3]])
  empty = tuple(t[:0] for t in problems[0][:5]) + problems[0][5:]
  got = emb.scans.lambda_returns([empty, problems[1]])
  assert got[0].shape[0] == 0
  np.testing.assert_allclose(got[1].cpu().numpy(), wants[1], **TOL)
  assert emb.scans.lambda_returns([]) == []


@pytest.mark.gpu
@pytest.mark.parametrize('seed,shape', TB)
def test_hip_director_score_matches_reference_fixture(emb, director, seed, shape):
  inp = cases.time_major(seed, shape)
  ret = emb.scans.director_score(dev(inp['rew']), dev(inp['cont']), dev(inp['value']),
                                 **cases.DIRECTOR_PARAMS)
  np.testing.assert_allclose(ret.cpu().numpy(), director['score_' + tag(seed, shape)], **TOL)


@pytest.mark.gpu
@pytest.mark.parametrize('seed,shape', TRAJ)
def test_hip_split_and_abstract_match_reference_fixture(emb, director, seed, shape):
  traj = cases.trajectory(seed, shape)
  k = cases.SKILL_DURATION
  cont = dev(traj['cont'])
  for key, x in traj.items():
    reward = key.startswith('reward_')
    got = emb.scans.split_traj(dev(x), k, reward).cpu().numpy()
    assert np.array_equal(got, director[f'split_{key}_' + tag(seed, shape)]), key   # pure movement
    kind = 'reward' if reward else 'cont' if key == 'cont' else 'first'
    got = emb.scans.abstract_traj(dev(x), cont, k, kind).cpu().numpy()
    np.testing.assert_allclose(got, director[f'abstract_{key}_' + tag(seed, shape)], **TOL)
