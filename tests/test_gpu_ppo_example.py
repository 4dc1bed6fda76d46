"""End to end: the example torch PPO agent (written against the Agent protocol)
learns CartPole through device Driver -> Replay(online) -> Consec -> GAE kernel."""
import importlib.util
import pathlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_ppo_cartpole_improves_over_random(tmp_path, monkeypatch):
  import embodied_amd as emb
  scores = []
  original = emb.utils.Logger.add

  def add(self, mapping, prefix=None):
    if prefix == 'episode':
      scores.append(float(mapping['score']))
    return original(self, mapping, prefix)

  monkeypatch.setattr(emb.utils.Logger, 'add', add)
  spec = importlib.util.spec_from_file_location('ppo_torch', ROOT / 'examples' / 'ppo_torch.py')
  module = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(module)
  logger = module.main(['--steps', '60000', '--logdir', str(tmp_path)])
  losses = [r['train/loss'] for r in logger.history if 'train/loss' in r]
  assert losses and np.isfinite(losses).all()
  early, late = np.mean(scores[:100]), np.mean(scores[-100:])
  assert early < 40                      # a random policy scores ~22
  assert late > 2 * early and late > 60, (early, late)
