"""bench.py's CPU pinning (pin_cpus): window choice per local rank, the knobs."""
import builtins
import importlib.util
import io
import os
import pathlib

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope='module')
def bench():
  spec = importlib.util.spec_from_file_location('_bench_for_pin_test', ROOT / 'bench.py')
  module = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(module)          # not __main__: nothing is pinned by the import
  assert module.PINNED is None
  return module


def _fake_host(monkeypatch, bench, busy_cpus, n=32):
  """A host with `n` allowed CPUs of which `busy_cpus` are fully busy."""
  calls = {'n': 0}
  real_open = builtins.open

  def stat():
    calls['n'] += 1
    tick = calls['n'] * 100
    lines = ['cpu  0 0 0 0 0 0 0 0']
    for c in range(n):
      user = tick if c in busy_cpus else 0
      idle = 0 if c in busy_cpus else tick
      lines.append(f'cpu{c} {user} 0 0 {idle} 0 0 0 0')
    return '\n'.join(lines) + '\n'

  def fake_open(path, *a, **k):
    if path == '/proc/stat':
      return io.StringIO(stat())
    return real_open(path, *a, **k)
  chosen = []
  monkeypatch.setattr(builtins, 'open', fake_open)
  monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(n)), raising=False)
  monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, cpus: chosen.append(list(cpus)), raising=False)
  monkeypatch.setattr(bench.time, 'sleep', lambda s: None)
  return chosen


def test_ranks_take_distinct_idle_windows(bench, monkeypatch):
  chosen = _fake_host(monkeypatch, bench, busy_cpus={0, 1, 2, 3, 9})
  monkeypatch.setattr(bench.sys, 'argv', ['bench.py'])
  picks = []
  for rank in range(4):
    monkeypatch.setenv('LOCAL_RANK', str(rank))
    picks.append(bench.pin_cpus())
  # windows 0-3 (all busy) and 8-11 (one busy CPU) come last; idle ones in order
  assert picks == [[4, 5, 6, 7], [12, 13, 14, 15], [16, 17, 18, 19], [20, 21, 22, 23]]
  assert chosen == picks


def test_pin_knobs(bench, monkeypatch):
  chosen = _fake_host(monkeypatch, bench, busy_cpus=set())
  monkeypatch.setattr(bench.sys, 'argv', ['bench.py', '--pin', '0'])
  assert bench.pin_cpus() is None and chosen == []
  monkeypatch.setattr(bench.sys, 'argv', ['bench.py', '--pin=10-13'])
  assert bench.pin_cpus() == [10, 11, 12, 13]
  monkeypatch.setattr(bench.sys, 'argv', ['bench.py', '--pin', 'auto'])
  monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(8)), raising=False)
  assert bench.pin_cpus() is None          # a small host (this container): left alone


def test_the_rank_launcher_does_not_pin_itself(bench, monkeypatch):
  monkeypatch.delenv('WORLD_SIZE', raising=False)
  monkeypatch.setattr(bench.sys, 'argv', ['bench.py', '--gpus', '8'])
  assert bench._is_launcher()
  monkeypatch.setattr(bench.sys, 'argv', ['bench.py', '--gpus=2', '--steps', '5'])
  assert bench._is_launcher()
  monkeypatch.setattr(bench.sys, 'argv', ['bench.py', '--gpus', '1'])
  assert not bench._is_launcher()
  monkeypatch.setenv('WORLD_SIZE', '8')
  monkeypatch.setattr(bench.sys, 'argv', ['bench.py', '--gpus', '8'])
  assert not bench._is_launcher()          # a rank under torch.distributed.run pins itself


def test_help_text_renders(bench, monkeypatch, capsys):
  """argparse formats every help string with %: a bare per-cent sign in one of
  them breaks `python bench.py --help` (and nothing else)."""
  monkeypatch.setattr('sys.argv', ['bench.py', '--help'])
  with pytest.raises(SystemExit) as stop:
    bench.parse()
  assert stop.value.code == 0
  assert '--comm' in capsys.readouterr().out
