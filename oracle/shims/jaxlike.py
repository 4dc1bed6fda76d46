"""numpy-backed stand-ins for the JAX names the reference's return-scan code
uses (`jnp`, `f32`, `sg`, `chex`).

TEST INFRASTRUCTURE, build container only: `oracle/gen_scan_golden.py`
executes the reference's own GAE / lambda-return / Director-score source text
(read from /root/reference at generation time, never written into this repo)
with these names bound, because JAX itself is not installed.

What this stands in for: XLA's float32 arithmetic.  Every operation the scan
code uses (+, -, *, slicing, stack, concatenate, cumprod, prod, mean, reshape,
transpose) is elementwise or a fixed-order reduction that numpy evaluates in
float32 as well; Python scalars are weakly typed on both sides (a Python float
times a float32 array stays float32, with the scalar rounded to float32).  The
one difference that can show is FMA contraction: XLA may fuse `a + b * c` into
one rounding where numpy rounds twice, ~1e-7 relative, far inside the 1e-5
budget of the parity tests.
"""
import numpy as np


class _Jnp:
  """`jax.numpy` as far as the scan code goes: numpy, float32 by default."""

  float32 = np.float32
  int32 = np.int32
  bool_ = np.bool_

  def __getattr__(self, name):
    return getattr(np, name)


jnp = _Jnp()
f32 = np.float32


def sg(x, skip=False):
  """stop_gradient: the identity on values."""
  return x


class _Chex:
  @staticmethod
  def assert_equal_shape(xs):
    shapes = {tuple(np.shape(x)) for x in xs}
    assert len(shapes) == 1, shapes


chex = _Chex()

NAMESPACE = {'jnp': jnp, 'f32': f32, 'sg': sg, 'chex': chex, 'np': np}
