"""The gradient size the multi-GPU bench all-reduces by default is the shipped PPO
model's parameter total, counted module by module (tools/count_params.py) -- not
an estimate.  The count is checked against a hand-derived breakdown and, in the
build container, against the reference's own yaml."""
import os
import pathlib
import re
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'tools'))


def test_ppo_parameter_total_is_the_default_gradient_size():
  import count_params as cp
  parts = cp.ppo(cp.PPO_AGENT, (84, 84, 4), 6)
  # impala encoder: three stages of (stage conv + two residual blocks of two 3x3 convs), 84 -> 42 -> 21 -> 11
  enc = ((3 * 3 * 4 * 32 + 32) + 4 * (3 * 3 * 32 * 32 + 32)
         + (3 * 3 * 32 * 64 + 64) + 4 * (3 * 3 * 64 * 64 + 64)
         + (3 * 3 * 64 * 64 + 64) + 4 * (3 * 3 * 64 * 64 + 64)
         + (11 * 11 * 64) * 512 + 512)
  assert parts['enc'] == enc == 4_354_464
  assert parts['actemb'] == 1024 + 6 * 1024 + 1024
  assert parts['rnn'] == 2 * (1024 + 512 + 1024) + (1024 + 512 + 1024) * 3072 + 3072      # layer norm + one Linear
  assert parts['policy'] == 1024 * 6 + 6 and parts['value'] == 1024 + 1
  total = sum(parts.values())
  assert total == 12_242_343
  bench = (ROOT / 'bench.py').read_text()
  assert re.search(r"--grad-numel', type=int, default=12_242_343\)", bench)
  assert re.search(r"--grad-dtype', default='f32'", bench)          # the reference's precision (jax/opt.py:52-54)


def test_counts_follow_the_reference_yaml_when_it_is_there():
  import count_params as cp
  ppo_cfg, dreamer_cfg = cp.from_reference('/root/reference')
  if ppo_cfg is None:            # (GPU box: no reference tree)
    return
  assert ppo_cfg == cp.PPO_AGENT and dreamer_cfg == cp.DREAMER_AGENT
  crafter = sum(cp.dreamer(dreamer_cfg, (64, 64, 3), 0, 17).values())
  assert 150e6 < crafter < 200e6          # "size200m" (dreamerv3/configs.yaml:146-149)
