"""Namespaces that present the reference, the oracle and the product through
one API so `tests/scenarios.py` can drive all three."""
import pathlib
import sys
import types

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
  sys.path.insert(0, str(ROOT))


def _tonp(x):
  if hasattr(x, 'detach'):
    return x.detach().cpu().numpy()
  return np.asarray(x)


def oracle_ns():
  from oracle import np_oracle as o
  ns = types.SimpleNamespace()
  ns.name = 'oracle'
  ns.Replay = o.Replay
  ns.Uniform = o.Uniform
  ns.Prioritized = o.Prioritized
  ns.Mixture = o.Mixture
  ns.SampleTree = o.SampleTree
  ns.Driver = o.Driver
  ns.tonp = _tonp
  ns.like = lambda template, array: np.asarray(array)
  ns.consec = lambda rep, batch, length, consec, prefix: iter(o.Consec(
      lambda: rep.sample(batch, 'train'), length, consec, prefix))
  return ns


def reference_ns():
  """Build container only: the real reference through oracle/refload.py."""
  from oracle import refload
  emb = refload.load()
  import elements
  ns = types.SimpleNamespace()
  ns.name = 'reference'

  def make_replay(**kw):
    elements.UUID.reset(debug=True)
    return emb.Replay(**kw)

  ns.Replay = make_replay
  ns.Uniform = emb.selectors.Uniform

  class TruthyPrioritized(emb.selectors.Prioritized):
    # replay.py:26 does `selector or Uniform(seed)`: an *empty* selector that
    # defines __len__ is falsy and silently replaced.  Force truthiness so the
    # reference's own Prioritized code is what the golden vectors record.
    def __bool__(self):
      return True

  ns.Prioritized = TruthyPrioritized
  ns.Recency = _repaired_recency(emb.selectors.Recency)
  ns.Mixture = emb.selectors.Mixture
  ns.SampleTree = emb.selectors.SampleTree
  ns.Driver = lambda envs: emb.Driver(
      [(lambda e=e: e) for e in envs], parallel=False)
  ns.wrappers = emb.core.wrappers
  ns.Space = elements.Space
  ns.tonp = _tonp
  ns.like = lambda template, array: np.asarray(array)
  ns.consec = lambda rep, batch, length, consec, prefix: iter(
      emb.streams.Consec(
          emb.streams.Stateless(rep.sample, batch, 'train'),
          length=length, consec=consec, prefix=prefix, strict=True,
          contiguous=True))
  return ns


def _repaired_recency(Recency):
  """The reference's Recency with ONE token of `_sample` repaired, made from its
  own source at run time (nothing of it is stored here).  selectors.py:98-105
  draws `rng.choice(len(segment), p=p)`: `segment` is unbound at the first level
  (UnboundLocalError on the first draw) and an int at the
  next; `p` -- the probability row the line already passes -- is what has a
  length.  Everything else (the table of `_build`, the age scaling, the item
  bookkeeping, the order of the generator's draws) is the reference's code."""
  import inspect
  import textwrap
  source = textwrap.dedent(inspect.getsource(Recency._sample))
  assert source.count('len(segment)') == 1, 'the reference changed: look at Recency._sample again'
  scope = {}
  exec(source.replace('len(segment)', 'len(p)'), {'np': np}, scope)
  return type('RepairedRecency', (Recency,), {'_sample': scope['_sample']})


def product_ns(device='cuda'):
  """The product: embodied_amd on its HIP library (needs a GPU)."""
  import torch
  import embodied_amd as emb
  ns = types.SimpleNamespace()
  ns.name = 'product'
  ns.Replay = lambda **kw: emb.Replay(device=device, **kw)
  ns.Uniform = emb.selectors.Uniform
  ns.Prioritized = emb.selectors.Prioritized
  ns.Recency = emb.selectors.Recency
  ns.Mixture = emb.selectors.Mixture
  ns.SampleTree = emb.selectors.SampleTree
  from embodied_amd.core import wrappers
  ns.wrappers = wrappers
  ns.Space = emb.Space
  ns.Driver = lambda envs: emb.Driver(
      [(lambda e=e: e) for e in envs], parallel=False, device=device)
  ns.tonp = _tonp

  def like(template, array):
    if hasattr(template, 'detach'):
      return torch.as_tensor(np.ascontiguousarray(array)).to(template.device)
    return np.asarray(array)

  ns.like = like
  ns.consec = lambda rep, batch, length, consec, prefix: iter(
      emb.streams.Consec(
          emb.streams.Stateless(rep.sample, batch, 'train'),
          length=length, consec=consec, prefix=prefix, strict=True,
          contiguous=True))
  return ns
