"""The C-ABI library loads and exports every symbol include/embodied_hip.h
declares (no compute calls: runs without a GPU)."""
import ctypes
import pathlib
import re

ROOT = pathlib.Path(__file__).resolve().parent.parent


def declared_symbols():
  text = (ROOT / 'include' / 'embodied_hip.h').read_text()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(emb_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
  from embodied_amd import _lib
  names = declared_symbols()
  assert len(names) > 50
  missing = [n for n in names if not hasattr(_lib.lib, n)]
  assert not missing, missing


def test_binding_covers_every_declared_symbol():
  from embodied_amd import _lib
  bound = set(_lib.SIGNATURES) | {'emb_last_error', 'emb_abi_version'}
  assert set(declared_symbols()) <= bound, set(declared_symbols()) - bound


def test_version_and_device_probe():
  from embodied_amd import _lib
  assert _lib.lib.emb_abi_version() == 5
  assert _lib.device_count() >= 0


def test_errors_are_reported_not_thrown():
  from embodied_amd import _lib
  import pytest
  with pytest.raises(ValueError):
    _lib.api.emb_tree_create(1, 0, ctypes.byref(ctypes.c_void_p()))   # branching < 2
  assert b'branching' in _lib.lib.emb_last_error()


def test_replay_refuses_cpu_device():
  import pytest
  import embodied_amd
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    embodied_amd.Replay(length=2, capacity=4, device='cpu')


def test_call_shim_reaches_the_same_exports():
  """csrc/fastcall.c calls the exported functions by address: same status
  handling as the ctypes binding, ints / None / ctypes arrays as arguments."""
  import ctypes as C
  import numpy as np
  import pytest
  from embodied_amd import _lib
  assert _lib.fast.module is not None, 'the call shim builds wherever gcc and Python.h exist'
  table = np.zeros(16, np.uint8)
  p = table.ctypes.data
  _lib.fast.emb_mask_actions(p, p, 0, 1, _lib.I32, p, None)           # n = 0: returns OK, no launch
  _lib.fast.emb_mask_actions((C.c_uint8 * 4)(), p, 0, 1, _lib.I32, p, None)
  with pytest.raises(ValueError, match='scan_gae'):
    _lib.fast.emb_scan_gae(0, 0, 0, 0, 0, 2, 0.5, 0.5, 0, 0, 0)       # null pointers: EMB_ERR_INVALID
  with pytest.raises(TypeError):
    _lib.fast.emb_mask_actions(C.byref(C.c_int()), p, 0, 1, _lib.I32, p, None)
  with pytest.raises(TypeError):
    _lib.fast.emb_mask_actions('x', p, 0, 1, _lib.I32, p, None)


def test_package_works_without_the_call_shim():
  import os
  import subprocess
  import sys
  code = ('from embodied_amd import _lib; assert _lib.fast.module is None; '
          'import numpy as np; t = np.zeros(4, np.uint8); p = t.ctypes.data; '
          '_lib.fast.emb_mask_actions(p, p, 0, 1, _lib.I32, p, None); print("ok")')
  env = dict(os.environ, EMB_NO_FASTCALL='1')
  out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  assert out.returncode == 0 and 'ok' in out.stdout, out.stderr


def test_bad_arguments_are_refused_not_dereferenced():
  """Null handles, null outputs, negative sizes, unknown keys: every entry point
  that can be reached without a GPU answers with a status code."""
  import ctypes as C
  import numpy as np
  from embodied_amd import _lib
  raw = _lib.lib
  for fn in (raw.emb_tree_sample, raw.emb_tree_len, raw.emb_tree_root_sum, raw.emb_selector_sample,
             raw.emb_selector_len, raw.emb_replay_len, raw.emb_replay_sampler_len,
             raw.emb_replay_free_slots):
    assert fn(None, None) < 0                                   # null handle
  tree, sel, rep = C.c_void_p(), C.c_void_p(), C.c_void_p()
  assert raw.emb_tree_create(16, 0, C.byref(tree)) == 0
  assert raw.emb_selector_create_prioritized(
      C.c_double(0.8), C.c_double(1.0), 1, C.c_double(0.5), 16, 0, C.byref(sel)) == 0
  cfg = _lib.ReplayConfig(4, 10, 8, 16, 0, 0, 0, 1, 0)
  assert raw.emb_replay_create(C.byref(cfg), None, C.c_uint64(0), C.byref(rep)) == 0
  assert raw.emb_tree_sample(tree, None) < 0 and raw.emb_tree_len(tree, None) < 0
  assert raw.emb_tree_root_sum(tree, None) < 0
  key = C.c_int64()
  assert raw.emb_tree_sample(tree, C.byref(key)) == _lib.ERR_EMPTY         # empty tree
  assert raw.emb_tree_remove(tree, C.c_int64(5)) == _lib.ERR_NOT_FOUND
  assert raw.emb_tree_update(tree, C.c_int64(5), C.c_double(1.0)) == _lib.ERR_NOT_FOUND
  assert raw.emb_selector_sample(sel, None) < 0 and raw.emb_selector_len(sel, None) < 0
  assert raw.emb_selector_sample(sel, C.byref(key)) == _lib.ERR_EMPTY
  assert raw.emb_selector_insert(sel, C.c_int64(1), None, 0) < 0            # prioritized item without steps
  assert raw.emb_selector_remove(sel, C.c_int64(77)) < 0
  assert raw.emb_selector_prioritize(sel, None, None, C.c_int64(3)) < 0
  assert raw.emb_replay_len(rep, None) < 0 and raw.emb_replay_stats(rep, None, 0) < 0
  rows = np.zeros(8, np.int32)
  assert raw.emb_replay_add_index(rep, C.c_int64(-1), None, None, None, None) < 0
  assert raw.emb_replay_add_index(rep, C.c_int64(2), None, C.c_void_p(rows.ctypes.data), None, None) < 0
  assert raw.emb_replay_sample_index(rep, C.c_int64(1), 0, C.c_void_p(rows.ctypes.data), None, None) == _lib.ERR_EMPTY
  assert raw.emb_replay_sample_index(rep, C.c_int64(1), 9, C.c_void_p(rows.ctypes.data), None, None) < 0   # bad mode
  assert raw.emb_replay_add(rep, C.c_int64(1), None, None, None) < 0
  assert raw.emb_replay_sample(rep, C.c_int64(1), 0, None, None, None, None) < 0
  assert raw.emb_replay_set_keys(rep, 0, None, None, None) < 0
  assert raw.emb_replay_grow(rep, C.c_int64(1), None) < 0                   # cannot shrink
  assert raw.emb_rng_create(None, 0, None) < 0
  assert raw.emb_np_sum(None, C.c_int64(3), None) < 0
  assert raw.emb_comm_init(None, 0, 1, None) < 0
  assert raw.emb_scan_gae(None, None, None, None, C.c_int64(1), C.c_int64(4), C.c_float(1), C.c_float(1),
                          None, None, None) < 0
  assert b'' != raw.emb_last_error()
  for destroy, handle in ((raw.emb_tree_destroy, tree), (raw.emb_selector_destroy, sel),
                          (raw.emb_replay_destroy, rep)):
    assert destroy(handle) == 0


def test_plain_c_program_against_the_abi(tmp_path):
  """examples/c_abi_demo.c: gcc -std=c99 against include/embodied_hip.h, linked
  with the shared library, no Python in the process.  It must print the index
  streams SURVEY.md Appendix C captured from the reference."""
  import os
  import pathlib
  import shutil
  import subprocess
  import pytest
  root = pathlib.Path(__file__).resolve().parent.parent
  gcc = shutil.which('gcc')
  if not gcc:
    pytest.skip('no gcc')
  exe = tmp_path / 'demo'
  build = subprocess.run(
      [gcc, '-std=c99', '-Wall', '-Werror', f'-I{root / "include"}', str(root / 'examples' / 'c_abi_demo.c'),
       f'-L{root / "embodied_amd"}', '-lembodied_hip', '-o', str(exe)], capture_output=True, text=True)
  assert build.returncode == 0, build.stderr
  import torch
  paths = [str(root / 'embodied_amd'), os.path.join(os.path.dirname(torch.__file__), 'lib'), '/opt/rocm/lib']
  env = dict(os.environ, LD_LIBRARY_PATH=':'.join(paths + [os.environ.get('LD_LIBRARY_PATH', '')]))
  run = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
  assert run.returncode == 0, run.stderr
  lines = run.stdout.strip().splitlines()
  assert lines[0] == 'uniform: 8 6 5 2 3 0 0 0 1 8 6 9 5 6 9 7'
  assert lines[1] == 'replay: 50 items; sampled workers 0 1 2 2'


def test_import_leaves_the_process_alone():
  """Importing the package changes neither the environment (the placement of
  kernel arguments, HIP_FORCE_DEV_KERNARG, is the host program's choice) nor the
  CPU affinity of the process."""
  import os
  import subprocess
  import sys
  code = ('import os\n'
          'before = (dict(os.environ), os.sched_getaffinity(0))\n'
          'import embodied_amd\n'
          'print(before == (dict(os.environ), os.sched_getaffinity(0)))')
  base = {k: v for k, v in os.environ.items() if k != 'HIP_FORCE_DEV_KERNARG'}
  for given in (None, '1', '0'):
    env = dict(base) if given is None else dict(base, HIP_FORCE_DEV_KERNARG=given)
    out = subprocess.run([sys.executable, '-c', code], env=env, cwd=str(ROOT),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == 'True', (given, out.stdout)


def test_knob_list_matches_the_knobs_the_sources_read():
  """csrc/knobs.h names every knob; emb_configure refuses anything else (a retired
  knob such as EMB_SPAN_VARIANT is an error, not a silent no-op)."""
  import pytest
  from embodied_amd import _lib
  src = ROOT / 'embodied_amd' / 'csrc'
  read = set()
  for path in src.iterdir():
    if path.suffix in ('.h', '.cpp', '.hip', '.c'):
      read |= set(re.findall(r'knob\("(EMB_[A-Z0-9_]+)"\)', path.read_text()))
  listed = set(re.findall(r'"(EMB_[A-Z0-9_]+)"', (src / 'knobs.h').read_text().split('kKnobNames[]')[1].split('};')[0]))
  assert read == listed, (read - listed, listed - read)
  documented = (ROOT / 'INTEGRATION.md').read_text()
  assert all(name in documented for name in listed), [n for n in listed if n not in documented]
  for name in ('EMB_SPAN_VARIANT', 'EMB_SAMPLE_POOL', 'EMB_NOT_A_KNOB'):
    with pytest.raises(ValueError, match='not a knob'):
      _lib.configure(**{name: 1})


def test_knobs_are_set_before_their_first_use_or_not_at_all():
  """emb_configure (include/embodied_hip.h): a knob is read once; the value a
  host program gives beats the environment; a late setting is refused instead of
  being half in effect.  Child process: knobs are process-wide."""
  import os
  import subprocess
  import sys
  code = '''
import os
os.environ["EMB_WHERE_BACKLOG"] = "65536"
import embodied_amd as emb
from embodied_amd import _lib
emb.configure(EMB_WHERE_BACKLOG=7, EMB_GATHER_STORES="plain")
emb.configure(EMB_GATHER_STORES=None)                      # withdrawn again
import numpy as np
sel = emb.selectors.Prioritized(exponent=0.8, initial=1.0, seed=0)
sel[0] = np.arange(60, dtype=np.uint8).reshape(3, 20)
sel[1] = np.arange(60, 120, dtype=np.uint8).reshape(3, 20)
del sel[0]                                                  # by now the selector has read EMB_WHERE_BACKLOG
try:
  emb.configure(EMB_WHERE_BACKLOG=9)
  print("late: accepted")
except ValueError as e:
  print("late: refused", "already in effect" in str(e))
try:
  emb.configure(SOMETHING=1)
  print("name: accepted")
except ValueError as e:
  print("name: refused")
# the knobs the Python layer reads go the same way (one accessor, _lib.knob)
os.environ["EMB_EARLY_INSERT"] = "1"
emb.configure(EMB_EARLY_INSERT=0)
from embodied_amd.core import driver as driverlib
from embodied_amd.envs import dummy
d = emb.Driver([lambda: dummy.Dummy("disc")], parallel=False)
print("python knob:", driverlib._EARLY_INSERT, driverlib._CARRY)
try:
  emb.configure(EMB_EARLY_INSERT=1)
  print("python late: accepted")
except ValueError as e:
  print("python late: refused", "already in effect" in str(e))
'''
  out = subprocess.run([sys.executable, '-c', code], cwd=str(ROOT), capture_output=True, text=True,
                       timeout=300)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = out.stdout.strip().splitlines()
  assert lines[-4:] == ['late: refused True', 'name: refused', 'python knob: False True',
                        'python late: refused True'], out.stdout
