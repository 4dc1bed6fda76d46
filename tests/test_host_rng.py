"""The C++ restatement of numpy's SeedSequence / PCG64 / Generator methods
(embodied_amd/csrc/np_random.h) against numpy itself.  Host only."""
import ctypes as C

import numpy as np
import pytest

from embodied_amd import _lib
from embodied_amd._lib import api


def words_of(seed):
  seeds = seed if isinstance(seed, (list, tuple)) else [seed]
  out = []
  for s in seeds:
    s = int(s)
    part = [s & 0xFFFFFFFF]
    s >>= 32
    while s:
      part.append(s & 0xFFFFFFFF)
      s >>= 32
    out += part
  return np.array(out, np.uint32)


class Rng:

  def __init__(self, seed):
    w = words_of(seed)
    self.h = C.c_void_p()
    api.emb_rng_create(_lib.ptr(w), len(w), C.byref(self.h))

  def integers(self, high, count=1):
    out = np.zeros(count, np.int64)
    api.emb_rng_integers(self.h, int(high), count, _lib.ptr(out))
    return out

  def random(self, count=1):
    out = np.zeros(count, np.float64)
    api.emb_rng_random(self.h, count, _lib.ptr(out))
    return out

  def choice(self, p, count=1):
    p = np.ascontiguousarray(p, np.float64)
    out = np.zeros(count, np.int64)
    api.emb_rng_choice(self.h, _lib.ptr(p), len(p), count, _lib.ptr(out))
    return out

  def __del__(self):
    api.raw.emb_rng_destroy(self.h)


@pytest.mark.parametrize('seed', [0, 1, 123, 2 ** 40 + 5, 2 ** 64 - 1, [7, 3], [0, 17]])
def test_seed_sequence_and_raw_stream(seed):
  ours, ref = Rng(seed), np.random.default_rng(seed)
  want = ref.integers(0, 2 ** 32, 64, dtype=np.uint32)
  # integers(0, 2**32) consumes one u32 per draw: same path as next_u32.
  got = ours.integers(2 ** 32, 64)
  assert (got == want).all()


@pytest.mark.parametrize('seed', [0, 5, 99])
def test_bounded_integers_mixed_sizes(seed):
  ours, ref = Rng(seed), np.random.default_rng(seed)
  sizes = np.random.default_rng(1000 + seed).integers(1, 5000, 600)
  sizes[::50] = 1                       # n == 1 draws nothing
  sizes[7::97] = 2 ** 31 + 12345        # heavy rejection region
  sizes[11::101] = 2 ** 32              # full 32-bit range
  sizes[13::103] = 2 ** 40 + 3          # 64-bit Lemire
  for n in sizes:
    assert ours.integers(int(n))[0] == ref.integers(0, int(n)).item(), n


def test_random_and_u32_buffer_interleave():
  ours, ref = Rng(42), np.random.default_rng(42)
  for i in range(300):
    if i % 3 == 0:
      assert ours.random()[0] == ref.random()
    else:
      assert ours.integers(1000)[0] == ref.integers(0, 1000).item()


@pytest.mark.parametrize('k', [1, 2, 3, 5, 16, 40])
def test_choice_with_probabilities(k):
  ours, ref = Rng(k), np.random.default_rng(k)
  gen = np.random.default_rng(77)
  for _ in range(50):
    p = gen.random(k)
    p[gen.random(k) < 0.2] = 0
    if p.sum() == 0:
      p[0] = 1
    p = p / p.sum()
    assert ours.choice(p)[0] == ref.choice(np.arange(k), p=p).item()


@pytest.mark.parametrize('n', list(range(0, 40)) + [127, 128, 129, 300, 1000])
def test_pairwise_sum_matches_ndarray_sum(n):
  gen = np.random.default_rng(n)
  for scale in (1.0, 1e-8, 1e12):
    x = (gen.random(n) - 0.3) * scale
    out = C.c_double()
    api.emb_np_sum(_lib.ptr(x), n, C.byref(out))
    assert out.value == x.sum()
