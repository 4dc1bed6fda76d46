"""End to end: the example torch PPO agent (written against the Agent protocol)
learns CartPole through device Driver -> Replay(online) -> Consec -> GAE kernel."""
import importlib.util
import pathlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_ppo_cartpole_improves_over_random(tmp_path):
  spec = importlib.util.spec_from_file_location('ppo_torch', ROOT / 'examples' / 'ppo_torch.py')
  module = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(module)
  logger = module.main(['--steps', '50000', '--logdir', str(tmp_path)])
  scores = [r['episode/score'] for r in logger.history if 'episode/score' in r]
  losses = [r['train/loss'] for r in logger.history if 'train/loss' in r]
  assert losses and np.isfinite(losses).all()
  assert scores and float(scores[-1]) > 45      # a random policy scores ~22
