"""The host concurrency core under ThreadSanitizer (tools/run_sanitizers.sh runs
the long form, with AddressSanitizer + UBSan beside it, into profiles/r05_*.txt).

csrc/replay_abi.cpp, index_abi.cpp and kernels_abi.cpp -- with replay_index.h,
selectors.h, defer_gate.h, stream_order.h -- are compiled by g++ against
tests/fake_hip (host memory as device memory, launches executed at once by plain
loops) and driven through the C ABI by tests/sanitize/soak.cpp: helper thread
on, early inserts on predicted rows, carried publishes, four sampler threads
that check every window against the generator, checkpoint-style bookkeeping.
A ~10 s slice here: a report, a failed window check or a stuck run fails it."""
import pathlib
import shutil
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
SOURCES = [
    ROOT / 'embodied_amd' / 'csrc' / 'replay_abi.cpp', ROOT / 'embodied_amd' / 'csrc' / 'index_abi.cpp',
    ROOT / 'embodied_amd' / 'csrc' / 'kernels_abi.cpp', ROOT / 'tests' / 'fake_hip' / 'fake_kernels.cpp',
    ROOT / 'tests' / 'sanitize' / 'soak.cpp']


@pytest.fixture(scope='module')
def soak(tmp_path_factory):
  gxx = shutil.which('g++')
  if not gxx:
    pytest.skip('no g++')
  binary = tmp_path_factory.mktemp('sanitize') / 'soak_tsan'
  cmd = [gxx, '-std=c++17', '-O1', '-g', '-fno-omit-frame-pointer', '-fsanitize=thread', '-pthread',
         f'-I{ROOT / "tests" / "fake_hip" / "include"}', f'-I{ROOT / "embodied_amd" / "csrc"}',
         *map(str, SOURCES), '-o', str(binary)]
  res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
  if res.returncode and 'tsan' in (res.stderr + res.stdout).lower():
    pytest.skip('this g++ has no ThreadSanitizer runtime')
  assert res.returncode == 0, res.stderr[-3000:]
  return binary


def _run(binary, *args):
  env = {'TSAN_OPTIONS': 'halt_on_error=0 history_size=4', 'PATH': '/usr/bin:/bin',
         'HIP_FORCE_DEV_KERNARG': '0'}           # bench.py's placement: argument rings in use
  res = subprocess.run([str(binary), *args], capture_output=True, text=True, timeout=300, env=env)
  text = res.stdout + res.stderr
  assert 'WARNING: ThreadSanitizer' not in text, text[-6000:]
  assert res.returncode == 0, text[-3000:]
  return text


@pytest.mark.parametrize('selector', ['uniform', 'prioritized'])
def test_stepping_loop_with_samplers_under_thread_sanitizer(soak, selector):
  text = _run(soak, '--seconds', '4', '--selector', selector, '--samplers', '4')
  assert 'errors 0' in text and 'windows checked' in text


def test_deferred_and_plain_paths_store_the_same_bytes_under_thread_sanitizer(soak):
  text = _run(soak, '--compare', '300', '--selector', 'prioritized')
  assert 'equal' in text and 'errors 0' in text
