"""Import `/root/reference/embodied/core` unmodified, in the build container only.

TEST INFRASTRUCTURE ONLY.  Used by `oracle/gen_golden.py` (and by the
container-only cross-check tests) to run the real reference against the numpy
restatement in `oracle/np_oracle.py`.  Nothing here is imported by the product
package, by `-m gpu` tests, by `smoke()` or by `bench.py`: `/root/reference`
does not exist on the GPU box.

The reference's top-level `embodied/__init__.py` imports `embodied.jax`, which
needs JAX (absent).  We therefore register an empty package object for
`embodied` whose `__path__` points at the reference tree, and import only
`embodied.core` and `embodied.envs.dummy` beneath it.
"""
import importlib
import os
import pathlib
import sys
import types

REFERENCE = pathlib.Path(os.environ.get('EMBODIED_REFERENCE', '/root/reference'))
SHIMS = pathlib.Path(__file__).parent / 'shims'


def available():
  return (REFERENCE / 'embodied' / 'core' / 'replay.py').exists()


def load():
  """Returns the reference `embodied` namespace (core symbols attached)."""
  if not available():
    raise RuntimeError(f'reference tree not found at {REFERENCE}')
  sys.dont_write_bytecode = True  # never drop __pycache__ into the mount
  if str(SHIMS) not in sys.path:
    sys.path.insert(0, str(SHIMS))
  if 'embodied' in sys.modules and getattr(
      sys.modules['embodied'], '_is_reference_stub', False):
    return sys.modules['embodied']
  pkg = types.ModuleType('embodied')
  pkg.__path__ = [str(REFERENCE / 'embodied')]
  pkg._is_reference_stub = True
  sys.modules['embodied'] = pkg
  core = importlib.import_module('embodied.core')
  for name in dir(core):
    if not name.startswith('_'):
      setattr(pkg, name, getattr(core, name))
  pkg.core = core
  pkg.replay = importlib.import_module('embodied.core.replay')
  envs = types.ModuleType('embodied.envs')
  envs.__path__ = [str(REFERENCE / 'embodied' / 'envs')]
  sys.modules['embodied.envs'] = envs
  pkg.envs = envs
  envs.dummy = importlib.import_module('embodied.envs.dummy')
  return pkg
