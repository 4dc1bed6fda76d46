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
  assert _lib.lib.emb_abi_version() == 1
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
