"""The return normaliser (embodied/jax/utils.py:16-91: running mean / std or
percentiles with debiasing, what `ppo_loss` and DreamerV3's `retnorm` scale
advantages and returns with) against a fixture made by EXECUTING the
reference's own class under numpy stand-ins (oracle/gen_normalize_golden.py).
CPU only.  One replica: what ranks exchange (`pmean`, the all-gather for the
percentiles) is checked against numpy in tests/test_distributed_gloo.py; this
pins the arithmetic every rank runs."""
import pathlib

import numpy as np
import pytest
import torch

from oracle import gen_normalize_golden as gen
from oracle import np_oracle

GOLDEN = pathlib.Path(__file__).parent / 'golden' / 'normalize.npz'
# float32 running statistics over 40 steps: the tolerance BASELINE.json's
# north_star states for float results
RTOL = ATOL = 1e-5


def golden():
  with np.load(GOLDEN) as f:
    return {k: f[k] for k in f.files}


@pytest.mark.parametrize('case', range(len(gen.CASES)))
def test_oracle_normalize_matches_the_reference_class(case):
  impl, fields = gen.CASES[case]
  want = golden()[f'case{case}']
  norm = np_oracle.Normalize(impl, **fields)
  rows = [np.array(norm.stats(), np.float64)]
  for step in range(gen.STEPS):
    norm.update([gen.inputs(case, step)])          # one replica: one part
    rows.append(np.array(norm.stats(), np.float64))
  # the restatement runs the same float32 operations in the same order
  assert np.array_equal(np.stack(rows), want)


@pytest.mark.parametrize('case', range(len(gen.CASES)))
def test_product_normalize_matches_the_reference_class(case):
  """`embodied_amd.distributed.Normalize` (torch; here on CPU tensors, no process
  group: `pmean` / the percentile gather are the identity with one rank)."""
  from embodied_amd import distributed as D
  impl, fields = gen.CASES[case]
  want = golden()[f'case{case}']
  norm = D.Normalize(impl, **fields)
  rows = [np.array([float(v) for v in norm.stats()], np.float64)]
  for step in range(gen.STEPS):
    stats = norm(torch.from_numpy(gen.inputs(case, step)), update=True)
    rows.append(np.array([float(v) for v in stats], np.float64))
  np.testing.assert_allclose(np.stack(rows), want, rtol=RTOL, atol=ATOL)


@pytest.mark.reference
def test_normalize_fixture_is_current():
  """Build container only: the committed fixture equals a fresh execution of the
  reference's class."""
  from oracle import refload
  if not refload.available():
    pytest.skip('no /root/reference here')
  Normalize, lines = gen.reference_class()
  want = golden()
  assert tuple(want['lines']) == lines
  for case, (impl, fields) in enumerate(gen.CASES):
    norm = Normalize(impl)
    for name, value in fields.items():
      setattr(norm, name, value)
    rows = [np.array(norm.stats(), np.float64)]
    for step in range(gen.STEPS):
      rows.append(np.array(norm(gen.inputs(case, step), True), np.float64))
    assert np.array_equal(np.stack(rows), want[f'case{case}']), case
