import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
  sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / 'tests' / 'golden'

# The HIP library is git-ignored: (re)build it in-tree if it is missing or older
# than its sources.  hipcc cross-compiles gfx950 without a GPU.
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location(
    '_emb_build', ROOT / 'embodied_amd' / 'build.py')   # not via the package:
_build = importlib.util.module_from_spec(_spec)         # its __init__ needs the .so
_spec.loader.exec_module(_build)
import os  # noqa: E402
os.environ.setdefault('EMB_STRICT_SCRATCH', '1')   # this repo's own builds: no kernel may spill
# The suite runs on the placement of kernel arguments that bench.py chooses (host
# memory; the package itself sets nothing).  The HIP runtime's own
# default (device memory, HIP_FORCE_DEV_KERNARG=1) is covered by child processes
# in tests/test_gpu_host_kernargs.py.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '0')
# Tests step slowly (an oracle step between two device steps): let the publish
# defer its index bookkeeping to the helper thread whatever the pace, so that the
# suite runs that path (tests/test_gpu_host_kernargs.py covers EMB_DEFER_INDEX=0).
os.environ.setdefault('EMB_DEFER_MAX_GAP_US', '1e9')
_build.build(verbose=False)


def pytest_configure(config):
  config.addinivalue_line(
      'markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')
  config.addinivalue_line(
      'markers', 'reference: needs /root/reference (build container only)')


def load_golden(name):
  with np.load(GOLDEN / f'{name}.npz') as f:
    return {k.replace('__', '/'): f[k] for k in f.files}


def assert_same(got, want, name=''):
  """Bit-exact for integer/byte/bool arrays, exact for floats that are pure
  copies (replay payloads)."""
  assert set(got) == set(want), (name, sorted(set(got) ^ set(want)))
  for key in want:
    a, b = np.asarray(got[key]), np.asarray(want[key])
    assert a.shape == b.shape, (name, key, a.shape, b.shape)
    assert a.dtype == b.dtype, (name, key, a.dtype, b.dtype)
    if not np.array_equal(a, b, equal_nan=a.dtype.kind == 'f'):
      bad = np.argwhere(a != b)[:5]
      raise AssertionError(f'{name}:{key} differs at {bad.tolist()}')


@pytest.fixture(scope='session')
def golden():
  return load_golden
