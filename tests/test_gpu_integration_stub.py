"""The reference-side ctypes stub printed in INTEGRATION.md is executed as it
stands (only the library path is pointed at the in-tree build) and must drive
add -> sample -> update -> GAE correctly."""
import pathlib
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def _stub_namespace():
  text = (ROOT / 'INTEGRATION.md').read_text()
  code = re.search(r'```python\n(# embodied/core/_hip\.py.*?)```', text, re.S).group(1)
  code = code.replace("C.CDLL('libembodied_hip.so')",
                      f"C.CDLL({str(ROOT / 'embodied_amd' / 'libembodied_hip.so')!r})")
  ns = {}
  exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
  return ns


def test_integration_stub_runs_the_path():
  ns = _stub_namespace()
  from oracle import np_oracle
  n, L, chunksize, slots = 3, 4, 8, 16
  h = ns['replay_create'](L, 20, chunksize, slots, False, seed=5)
  specs = [('obs', torch.float32, (6,)), ('is_first', torch.bool, ()), ('is_last', torch.bool, ()),
           ('stepid', torch.uint8, (20,))]
  rowbytes = [int(np.prod(s, dtype=np.int64)) * torch.empty((), dtype=d).element_size() for _, d, s in specs]
  pools = [torch.zeros(slots * chunksize * rb, dtype=torch.uint8, device='cuda') for rb in rowbytes]
  ns['replay_set_keys'](h, [k for k, _, _ in specs], rowbytes, [p.data_ptr() for p in pools])
  ref = np_oracle.Replay(L, 20, chunksize, seed=5)
  stream = torch.cuda.current_stream().cuda_stream
  for t in range(15):
    obs = torch.arange(n * 6, dtype=torch.float32, device='cuda').reshape(n, 6) + 100 * t
    first = torch.full((n,), t % 7 == 0, device='cuda')
    last = torch.full((n,), t % 7 == 6, device='cuda')
    ns['replay_add'](h, list(range(n)), [obs.data_ptr(), first.data_ptr(), last.data_ptr(), 0], stream)
    for w in range(n):
      ref.add({'obs': obs[w].cpu().numpy(), 'is_first': bool(first[w]), 'is_last': bool(last[w])}, w)
    torch.cuda.synchronize()
  B = 5
  outs = [torch.empty((B, L, *s), dtype=d, device='cuda') for _, d, s in specs]
  ns['replay_sample'](h, B, 'train', [o.data_ptr() for o in outs], stream)
  want = ref.sample(B)
  for (name, _, _), got in zip(specs, outs):
    assert np.array_equal(got.cpu().numpy(), want[name]), name
  # write new `obs` over the first 3 steps of every sampled window, read it back
  new = torch.randn((B, 3, 6), device='cuda')
  first_ids = np.ascontiguousarray(outs[3][:, 0].cpu().numpy())
  ns['replay_update'](h, B, 3, first_ids, [0], [new.data_ptr()], stream)
  ref.update({'stepid': outs[3][:, :3].cpu().numpy(), 'obs': new.cpu().numpy()})
  again = [torch.empty((B, L, *s), dtype=d, device='cuda') for _, d, s in specs]
  ns['replay_sample'](h, B, 'train', [o.data_ptr() for o in again], stream)
  want = ref.sample(B)
  assert np.array_equal(again[0].cpu().numpy(), want['obs'])
  # GAE through the stub
  rew = torch.randn(4, 9, device='cuda'); val = torch.randn(4, 9, device='cuda')
  flags = torch.zeros(4, 9, dtype=torch.bool, device='cuda')
  adv = torch.empty(4, 8, device='cuda'); tar = torch.empty(4, 8, device='cuda')
  ns['scan_gae'](rew.data_ptr(), val.data_ptr(), flags.data_ptr(), flags.data_ptr(), 4, 9, 200, 0.8,
                 adv.data_ptr(), tar.data_ptr(), stream)
  wadv, wtar = np_oracle.gae(rew.cpu().numpy(), val.cpu().numpy(), flags.cpu().numpy(),
                             flags.cpu().numpy(), hor=200, lam=0.8)
  np.testing.assert_allclose(adv.cpu().numpy(), wadv, rtol=1e-5, atol=1e-5)
  # masked actions and obs stack entry points
  act = torch.tensor([1.5, -2.0, 3.0], device='cuda'); out = torch.empty_like(act)
  ended = torch.tensor([False, True, False], device='cuda')
  ns['mask_actions'](act.data_ptr(), out.data_ptr(), 3, 1, 7, ended.data_ptr(), stream)   # 7 = EMB_F32
  assert out.tolist() == [1.5, -0.0, 3.0] and np.signbit(out.cpu().numpy()[1])
  frames = torch.randint(0, 255, (2, 4, 4, 3), dtype=torch.uint8, device='cuda')
  stacked = torch.empty((2, 3, 4, 4), dtype=torch.float32, device='cuda')
  ns['obs_stack'](frames.data_ptr(), 2, 16, 3, stacked.data_ptr(), 7, 1 / 255, stream)
  torch.testing.assert_close(stacked, frames.permute(0, 3, 1, 2).float() / 255)
  ns['lib'].emb_replay_destroy(h)
  # The early insert: observation keys with the obs stack, the action after the
  # policy -- against the oracle's plain Replay.add of the same steps.
  n, L, chunksize, slots = 3, 2, 8, 16
  h = ns['replay_create'](L, 40, chunksize, slots, False, seed=1)
  specs = [('image', torch.uint8, (4, 4, 4)), ('is_first', torch.bool, ()), ('is_last', torch.bool, ()),
           ('action', torch.float32, ()), ('stepid', torch.uint8, (20,))]
  rowbytes = [int(np.prod(s, dtype=np.int64)) * torch.empty((), dtype=d).element_size() for _, d, s in specs]
  pools = [torch.zeros(slots * chunksize * rb, dtype=torch.uint8, device='cuda') for rb in rowbytes]
  ns['replay_set_keys'](h, [k for k, _, _ in specs], rowbytes, [p.data_ptr() for p in pools])
  ref = np_oracle.Replay(L, 40, chunksize, seed=1)
  spec = ns['ObsSpec'](16, 4, 1, 7, 1 / 255, 0.0)          # (N, 16, 4) u8 -> (N, 4, 16) f32 / 255
  tokens = []
  for t in range(12):
    image = torch.randint(0, 255, (n, 4, 4, 4), dtype=torch.uint8, device='cuda')
    first = torch.full((n,), t == 0, device='cuda')
    last = torch.tensor([t % 5 == 4, False, t % 3 == 2], device='cuda')
    batch = torch.empty((n, 4, 4, 4), dtype=torch.float32, device='cuda')
    token = ns['obs_stack_insert'](h, list(range(n)), 0, image.data_ptr(), spec, batch.data_ptr(),
                                   [image.data_ptr(), first.data_ptr(), last.data_ptr(), 0, 0], stream)
    tokens.append(token)
    torch.testing.assert_close(batch, image.permute(0, 3, 1, 2).float() / 255)
    action = torch.full((n,), -1.0 - t, device='cuda')       # the policy's answer
    masked = torch.empty_like(action)
    ns['replay_publish'](h, list(range(n)), [image.data_ptr(), first.data_ptr(), last.data_ptr(),
                                             action.data_ptr(), 0],
                         [3], [7], [masked.data_ptr()], last.data_ptr(), token, stream)
    torch.cuda.synchronize()
    want_masked = action.cpu().numpy() * ~last.cpu().numpy()
    assert np.array_equal(masked.cpu().numpy(), want_masked) and np.array_equal(
        np.signbit(masked.cpu().numpy()), np.signbit(want_masked))
    for w in range(n):
      ref.add({'image': image[w].cpu().numpy(), 'is_first': bool(first[w]), 'is_last': bool(last[w]),
               'action': np.float32(want_masked[w])}, w)
  assert tokens[0] == 0 and all(tokens[1:])      # the first step opens the workers' chunks
  outs = [torch.empty((6, L, *s), dtype=d, device='cuda') for _, d, s in specs]
  ns['replay_sample'](h, 6, 'train', [o.data_ptr() for o in outs], stream)
  want = ref.sample(6)
  for (name, _, _), got in zip(specs, outs):
    assert np.array_equal(got.cpu().numpy(), want[name]), name
  ns['lib'].emb_replay_destroy(h)
