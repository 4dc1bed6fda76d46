"""Pin the CPU oracle against golden vectors recorded from the real reference
(oracle/gen_golden.py).  CPU only."""
import pytest

from tests import adapters, scenarios
from tests.conftest import assert_same, load_golden


@pytest.mark.parametrize('name', sorted(scenarios.SCENARIOS))
def test_oracle_matches_reference_golden(name):
  got = scenarios.SCENARIOS[name](adapters.oracle_ns())
  assert_same(got, load_golden(name), name)


@pytest.mark.reference
@pytest.mark.parametrize('name', sorted(scenarios.SCENARIOS))
def test_golden_is_current(name):
  """Build container only: the committed fixtures equal a fresh reference run."""
  from oracle import refload
  if not refload.available():
    pytest.skip('no /root/reference here')
  got = scenarios.SCENARIOS[name](adapters.reference_ns())
  assert_same(got, load_golden(name), name)


def test_survey_appendix_c_values():
  """The values SURVEY.md Appendix C captured from the reference."""
  g = load_golden('replay_basic')
  assert g['lens'][-1] == 50
  assert g['s0/step'][:, 0].tolist() == [14, 10, 24, 20]
  assert g['s0/worker'][:, 0].tolist() == [0, 1, 2, 2]
  assert g['s0/is_first'].astype(int).tolist() == [
      [1, 0, 0, 0, 0], [1, 1, 0, 0, 0], [1, 0, 0, 0, 0], [1, 0, 1, 0, 0]]
  assert g['s0/is_last'].astype(int).tolist() == [
      [0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 1, 0, 0, 0]]
  u = load_golden('sel_uniform')
  assert u['draws_a'].tolist() == [8, 6, 5, 2, 3, 0, 0, 0, 1, 8, 6, 9, 5, 6, 9, 7]
  assert u['draws_b'].tolist() == [5, 4, 5, 8, 2, 7, 6, 0]
