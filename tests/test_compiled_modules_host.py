"""The Cython-compiled copies of the hot host modules (embodied_amd/_compiled,
build.py `build_compiled`): optional, never stale, switchable."""
import hashlib
import json
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_finder_only_serves_binaries_whose_source_digest_matches(tmp_path, monkeypatch):
  from embodied_amd import _compiled_finder as F
  monkeypatch.delenv('EMB_PURE_PYTHON', raising=False)
  root, out = tmp_path / 'src', tmp_path / 'bin'
  (root / 'pkg').mkdir(parents=True)
  out.mkdir()
  (root / 'pkg' / 'fresh.py').write_text('x = 1\n')
  (root / 'pkg' / 'edited.py').write_text('x = 2  # changed after the build\n')
  (root / 'pkg' / 'gone.py').write_text('x = 3\n')
  for name in ('pkg.fresh.so', 'pkg.edited.so'):
    (out / name).write_bytes(b'\x7fELF')
  digest = lambda text: hashlib.sha256(text.encode()).hexdigest()
  (out / 'manifest.json').write_text(json.dumps({
      'pkg.fresh': {'file': 'pkg.fresh.so', 'source': 'pkg/fresh.py', 'sha256': digest('x = 1\n')},
      'pkg.edited': {'file': 'pkg.edited.so', 'source': 'pkg/edited.py', 'sha256': digest('x = 2\n')},
      'pkg.gone': {'file': 'pkg.gone.so', 'source': 'pkg/gone.py', 'sha256': digest('x = 3\n')},
      'pkg.nosource': {'file': 'pkg.fresh.so', 'source': 'pkg/nosource.py', 'sha256': '0'},
  }))
  finder = F.Finder(out, root)
  assert set(finder.table) == {'pkg.fresh'}          # edited source, missing binary, missing source: skipped
  spec = finder.find_spec('pkg.fresh')
  assert spec.origin.endswith('pkg.fresh.so') and finder.find_spec('pkg.edited') is None
  monkeypatch.setenv('EMB_PURE_PYTHON', '1')
  assert F.Finder(out, root).table == {}
  (out / 'manifest.json').write_text('not json')
  monkeypatch.delenv('EMB_PURE_PYTHON')
  assert F.Finder(out, root).table == {}


def _loaded(env):
  code = ('import embodied_amd as e, embodied_amd.core.driver as d, json; '
          'print(json.dumps([sorted(e.compiled.loaded), d.__file__]))')
  res = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=300)
  assert res.returncode == 0, res.stderr[-2000:]
  return json.loads(res.stdout.strip().splitlines()[-1])


def test_pure_python_switch_and_compiled_default():
  names, origin = _loaded({'EMB_PURE_PYTHON': '1'})
  assert names == [] and origin.endswith('core/driver.py')
  manifest = ROOT / 'embodied_amd' / '_compiled' / 'manifest.json'
  if not manifest.exists():
    pytest.skip('host modules not compiled here (python embodied_amd/build.py)')
  spec = __import__('importlib.util').util.spec_from_file_location(
      '_emb_build_probe', ROOT / 'embodied_amd' / 'build.py')
  builder = __import__('importlib.util').util.module_from_spec(spec)
  spec.loader.exec_module(builder)
  if builder.compiled_stale():
    pytest.skip('compiled host modules are older than their sources: the plain ones are in use')
  names, origin = _loaded({'EMB_PURE_PYTHON': '0'})
  assert set(names) >= {'embodied_amd.core.driver', 'embodied_amd.core.replay'}
  assert '_compiled' in origin
