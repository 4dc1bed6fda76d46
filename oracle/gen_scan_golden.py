"""Generate tests/golden/scan_*.npz by EXECUTING the reference's own return-scan
source under numpy stand-ins for the JAX names it uses.

TEST INFRASTRUCTURE, build container only (needs /root/reference).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_scan_golden.py

The reference's scan code is JAX (`ppo/agent.py:177-232 ppo_loss`,
`dreamerv3/agent.py:482-490 lambda_return`, `director/agent.py:430-445
VFunction.score`, `director/hierarchy.py:224-256 Director.split_traj /
abstract_traj`) and JAX is not installed, so the modules cannot be imported.
The functions themselves are plain array arithmetic with a Python `for t in
reversed(range(...))` loop: this script parses each file at generation time,
takes the one function definition out of the syntax tree, compiles THAT (the
reference's text is never written anywhere) and calls it with `jnp`, `f32`,
`sg`, `chex` bound to `oracle/shims/jaxlike.py` and small recording stand-ins
for the objects around it (policy / value heads, normalisers, `self`).  Only
data — seeded inputs' digests and the functions' outputs — is written.

numpy stands in for XLA's float32 arithmetic (see oracle/shims/jaxlike.py: FMA
contraction may differ at ~1e-7).
"""
import ast
import pathlib
import sys
import types

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True

from oracle import refload  # noqa: E402
from oracle.shims import jaxlike  # noqa: E402
from tests import scan_cases as cases  # noqa: E402


def extract(relpath, name, cls=None):
  """The function `name` (inside class `cls`, if given) of a reference file,
  compiled on its own with the stand-in names as its globals."""
  path = refload.REFERENCE / relpath
  tree = ast.parse(path.read_text(), filename=str(path))
  body = tree.body
  if cls is not None:
    body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
  node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
  module = ast.Module(body=[node], type_ignores=[])
  namespace = dict(jaxlike.NAMESPACE)
  exec(compile(module, str(path), 'exec'), namespace)
  return namespace[name], (node.lineno, node.end_lineno)


class Recorder:
  """A normaliser stand-in: offset 0, scale 1, remembers what it was fed."""

  def __init__(self):
    self.seen = None

  def stats(self):
    return 0.0, 1.0

  def __call__(self, x, update):
    self.seen = np.asarray(x)
    return 0.0, 1.0


def run_ppo_loss(ppo_loss, inp, hor, lam):
  """ppo/agent.py ppo_loss, whole and unmodified: `adv` and `tar` are read off
  the two normalisers the function hands them to."""
  shape = inp['rew'].shape
  zeros = np.zeros(shape, np.float32)
  head = types.SimpleNamespace(logp=lambda a: zeros, entropy=lambda: zeros)
  value = types.SimpleNamespace(pred=lambda: inp['val'], loss=lambda target: zeros)
  advnorm, valnorm = Recorder(), Recorder()
  data = {
      'reward': inp['rew'], 'is_last': inp['last'], 'is_terminal': inp['term'],
      'action': np.zeros(shape, np.int32), 'logp/action': zeros}
  ppo_loss(data, {'action': head}, value, advnorm, valnorm, {'action': None}, True,
           hor=hor, lam=lam)
  return advnorm.seen, valnorm.seen


def main():
  outdir = ROOT / 'tests' / 'golden'
  outdir.mkdir(exist_ok=True)
  ppo_loss, ppo_lines = extract('ppo/agent.py', 'ppo_loss')
  lambda_return, lam_lines = extract('dreamerv3/agent.py', 'lambda_return')
  score, score_lines = extract('director/agent.py', 'score', cls='VFunction')
  split_traj, split_lines = extract('director/hierarchy.py', 'split_traj', cls='Director')
  abstract_traj, abs_lines = extract('director/hierarchy.py', 'abstract_traj', cls='Director')
  print('reference functions:', dict(
      ppo_loss=ppo_lines, lambda_return=lam_lines, score=score_lines,
      split_traj=split_lines, abstract_traj=abs_lines))

  out = {}
  for seed in cases.SEEDS:
    for shape in cases.SHAPES_BT:
      inp = cases.batch_major(seed, shape)
      tag = f's{seed}_{shape[0]}x{shape[1]}'
      out[f'in_{tag}'] = cases.digest(inp)
      adv, tar = run_ppo_loss(ppo_loss, inp, **cases.GAE_PARAMS)
      assert adv.dtype == tar.dtype == np.float32, (adv.dtype, tar.dtype)
      out[f'gae_adv_{tag}'], out[f'gae_tar_{tag}'] = adv, tar
      for i, params in enumerate(cases.LAMBDA_PARAMS):
        if i and seed > 1:
          continue
        ret = lambda_return(inp['last'], inp['term'], inp['rew'], inp['val'], inp['boot'],
                            params['disc'], params['lam'])
        assert ret.dtype == np.float32, ret.dtype
        out[f'lambda{i}_{tag}'] = ret
  np.savez_compressed(outdir / 'scan_batch_major.npz', **out)
  print('scan_batch_major', len(out), 'arrays', (outdir / 'scan_batch_major.npz').stat().st_size, 'bytes')

  out = {}
  for seed in cases.SEEDS:
    for shape in cases.SHAPES_TB:
      inp = cases.time_major(seed, shape)
      tag = f's{seed}_{shape[0]}x{shape[1]}'
      out[f'in_{tag}'] = cases.digest(inp)
      critic = types.SimpleNamespace(
          rewfn=lambda traj: traj['reward'],
          config=types.SimpleNamespace(
              horizon=cases.DIRECTOR_PARAMS['horizon'], return_lambda=cases.DIRECTOR_PARAMS['lam']),
          net=lambda traj: types.SimpleNamespace(mean=lambda: traj['value']))
      traj = {'cont': inp['cont'], 'reward': inp['rew'], 'value': inp['value']}
      rew, ret, base = score(critic, traj)
      assert ret.dtype == np.float32 and ret.shape == inp['rew'].shape
      out[f'score_{tag}'] = ret
  director = types.SimpleNamespace(
      config=types.SimpleNamespace(train_skill_duration=cases.SKILL_DURATION))
  for seed in cases.SEEDS[:2]:
    for shape in cases.TRAJ_SHAPES_TB:
      traj = cases.trajectory(seed, shape)
      tag = f's{seed}_{shape[0]}x{shape[1]}'
      out[f'trajin_{tag}'] = cases.digest(traj)
      for key, value in split_traj(director, traj).items():
        out[f'split_{key}_{tag}'] = np.ascontiguousarray(value)
      for key, value in abstract_traj(director, traj).items():
        out[f'abstract_{key}_{tag}'] = np.ascontiguousarray(value)
  np.savez_compressed(outdir / 'scan_director.npz', **out)
  print('scan_director', len(out), 'arrays', (outdir / 'scan_director.npz').stat().st_size, 'bytes')


if __name__ == '__main__':
  main()
