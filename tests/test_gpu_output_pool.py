"""Who owns a sampled batch.  `Replay.sample` returns tensors the caller owns,
like the reference's fresh arrays (embodied/core/replay.py:255-275).  Reuse of
output tensors is explicit -- `Replay.recycle`, `sample(out=)`,
`streams.Stateless(recycle=K)`, `scans.gae(out=)`, `Replay(reuse_outputs=K)`:
nothing is found out behind the caller's back."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def emb():
  import embodied_amd
  assert torch.cuda.is_available(), 'these tests need the MI355X'
  return embodied_amd


def _filled(emb, **kw):
  rep = emb.Replay(length=4, capacity=200, chunksize=16, seed=0, **kw)
  for t in range(60):
    rep.add({'x': np.float32(t), 'is_first': t == 0, 'is_last': False}, 0)
  return rep


def test_fresh_tensors_by_default(emb):
  rep = _filled(emb)
  seen, kept = set(), []
  for _ in range(6):
    batch = rep.sample(5)
    assert isinstance(batch, dict) and set(batch) == {'x', 'is_first', 'is_last', 'stepid'}
    kept.append((batch, {k: v.clone() for k, v in batch.items()}))
    seen.add(batch['x'].data_ptr())
  assert len(seen) == 6                                   # held batches never share storage
  for batch, snapshot in kept:
    for k in snapshot:
      assert torch.equal(batch[k], snapshot[k]), k


def test_recycle_hands_a_set_out_again_and_only_that_set(emb):
  rep, twin = _filled(emb), _filled(emb)
  first = rep.sample(5)
  twin.sample(5)
  ptr = first['x'].data_ptr()
  held = rep.sample(5)
  want_held = twin.sample(5)
  rep.recycle(first)
  rep.recycle(first)                                      # twice: still one entry
  again = rep.sample(5)
  want = twin.sample(5)
  assert again['x'].data_ptr() == ptr and again is first  # the very set that was handed back
  assert rep.sample(5)['x'].data_ptr() not in (ptr, held['x'].data_ptr())
  twin.sample(5)
  for k in want:
    assert torch.equal(again[k], want[k]) and torch.equal(held[k], want_held[k]), k
  # update() through a recycled set uses THIS batch's first-step ids
  rep.update({'stepid': again['stepid'], 'x': torch.full_like(again['x'], -7.0)})
  rows, _ = rep.sample_index(5)
  with pytest.raises(TypeError):
    rep.recycle(dict(again))                              # not a batch this replay made
  with pytest.raises(TypeError):
    _filled(emb).recycle(again)                           # another replay's
  # gather() into a recycled set drops the host copy of an older batch's step ids
  rep.recycle(again)
  batch = rep.gather(rows)
  assert batch is again and getattr(batch['stepid'], '_emb_first', None) is None
  rep.update({'stepid': batch['stepid'], 'x': torch.full_like(batch['x'], -9.0)})
  assert (rep.gather(rows)['x'] == -9.0).all()


def test_sample_into_caller_owned_tensors(emb):
  rep, twin = _filled(emb), _filled(emb)
  own = {k: torch.empty_like(v) for k, v in rep.sample(5).items()}
  twin.sample(5)
  ptrs = {k: v.data_ptr() for k, v in own.items()}
  got = rep.sample(5, out=own)
  want = twin.sample(5)
  for k in want:
    assert got[k].data_ptr() == ptrs[k] and torch.equal(own[k], want[k]), k
  again = rep.sample(5, out=got)                          # a Batch of this replay: taken as is
  want = twin.sample(5)
  assert again is got and all(torch.equal(again[k], want[k]) for k in want)
  with pytest.raises(ValueError):
    rep.sample(6, out=own)                                # wrong batch size
  with pytest.raises(KeyError):
    rep.sample(5, out={'x': own['x']})
  bad = dict(own, x=own['x'].double())
  with pytest.raises(ValueError):
    rep.sample(5, out=bad)
  # a rejected `out` consumed no draw
  assert all(torch.equal(rep.sample(5)[k], twin.sample(5)[k]) for k in want)


def test_stateless_recycle_lends_batches_for_k_draws(emb):
  rep, twin = _filled(emb), _filled(emb)
  K = 2
  stream = emb.streams.Stateless(rep.sample, 5, 'train', recycle=K)
  plain = emb.streams.Stateless(twin.sample, 5, 'train')
  history, ptrs = [], []
  for i in range(12):
    batch, want = next(stream), next(plain)
    ptrs.append(batch['x'].data_ptr())
    history.append((batch, {k: v.clone() for k, v in want.items()}))
    for old, snapshot in history[-(K + 1):]:              # lent for K further draws: unchanged
      for k in snapshot:
        assert torch.equal(old[k], snapshot[k]), (i, k)
  assert len(set(ptrs)) == K + 1                          # K + 1 sets circulate, nothing else is allocated
  with pytest.raises(TypeError):
    emb.streams.Stateless(lambda: 0, recycle=1)
  with pytest.raises(ValueError):
    emb.streams.Prefetch(emb.streams.Stateless(rep.sample, 5, recycle=2), amount=1)
  ahead = iter(emb.streams.Prefetch(emb.streams.Stateless(rep.sample, 5, recycle=3), amount=1))
  seen = [next(ahead) for _ in range(8)]
  assert len({b['x'].data_ptr() for b in seen}) <= 4
  # Consec (identity window) over a lending source serves the same tensors
  windows = iter(emb.streams.Consec(emb.streams.Stateless(rep.sample, 5, recycle=1), length=3, consec=1, prefix=1))
  assert len({next(windows)['x'].data_ptr() for _ in range(8)}) == 2


def test_gae_into_caller_owned_tensors(emb):
  from oracle import np_oracle
  gen = np.random.default_rng(0)
  rew, val = (gen.standard_normal((16, 64)).astype(np.float32) for _ in range(2))
  last, term = gen.random((16, 64)) < 0.02, gen.random((16, 64)) < 0.01
  args = [torch.as_tensor(a).cuda() for a in (rew, val, last, term)]
  adv, tar = torch.empty(16, 63, device='cuda'), torch.empty(16, 63, device='cuda')
  got = emb.scans.gae(*args, out=(adv, tar))
  assert got[0] is adv and got[1] is tar
  want = np_oracle.gae(rew, val, last, term)
  np.testing.assert_allclose(adv.cpu().numpy(), want[0], rtol=1e-5, atol=1e-5)
  np.testing.assert_allclose(tar.cpu().numpy(), want[1], rtol=1e-5, atol=1e-5)
  fresh = [emb.scans.gae(*args) for _ in range(3)]        # default: results never share storage
  assert len({a.data_ptr() for a, _ in fresh}) == 3
  with pytest.raises(ValueError):
    emb.scans.gae(*args, out=(adv[:, :5], tar))
