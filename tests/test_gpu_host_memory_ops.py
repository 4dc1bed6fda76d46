"""The two kernels of round 6 that touch pinned HOST memory directly (the Driver's
real-simulator path, embodied/core/driver.py:17-25,61-65,72-75 on the other side of
PCIe): `emb_copy_bytes` reads a pinned block and writes device memory, at any size
and alignment; `emb_mask_actions_notify` stores value * ~is_last into pinned memory
and then a sequence word the host polls -- from launches of one and of many
workgroups, back to back on one counter."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def emb():
  import embodied_amd
  assert torch.cuda.is_available(), 'these tests need the MI355X'
  return embodied_amd


@pytest.mark.parametrize('nbytes,src_off,dst_off', [
    (1, 0, 0), (15, 0, 0), (16, 0, 0), (4097, 0, 0), (28224 * 16, 0, 0), (1806336, 0, 0),
    (1000, 3, 0), (1000, 0, 5), (65536 + 7, 16, 32), (3 << 20, 1, 1)])
def test_copy_bytes_from_pinned_host_memory(emb, nbytes, src_off, dst_off):
  from embodied_amd import _lib
  gen = np.random.default_rng(nbytes)
  host = torch.from_numpy(gen.integers(0, 256, nbytes + 64, dtype=np.uint8)).pin_memory()
  dev = torch.full((nbytes + 128,), 0xEE, dtype=torch.uint8, device='cuda')
  _lib.fast.emb_copy_bytes(host.data_ptr() + src_off, dev.data_ptr() + dst_off, nbytes, _lib.raw_stream(dev.device))
  got = dev.cpu().numpy()
  assert np.array_equal(got[dst_off: dst_off + nbytes], host.numpy()[src_off: src_off + nbytes])
  assert (got[:dst_off] == 0xEE).all() and (got[dst_off + nbytes:] == 0xEE).all()       # nothing outside
  # ... and device to device
  twin = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
  _lib.api.emb_copy_bytes(dev.data_ptr() + dst_off, twin.data_ptr(), nbytes, _lib.raw_stream(dev.device))
  assert np.array_equal(twin.cpu().numpy(), host.numpy()[src_off: src_off + nbytes])
  _lib.api.emb_copy_bytes(None, None, 0, None)                                            # nothing to do: fine
  with pytest.raises(ValueError):
    _lib.api.emb_copy_bytes(None, twin.data_ptr(), 8, None)


@pytest.mark.parametrize('n,row_elems,dtype', [
    (64, 1, torch.int32), (64, 6, torch.float32), (7, 3, torch.bfloat16), (64, 17, torch.float64),
    (4096, 33, torch.float32)])               # the last one: 528 workgroups on one counter
def test_mask_actions_notify_writes_rows_then_the_word(emb, n, row_elems, dtype):
  from embodied_amd import _lib
  from embodied_amd.core.driver import _DTYPE_CODE, mask_actions
  gen = torch.Generator(device='cuda')
  gen.manual_seed(n * 31 + row_elems)
  is_last = torch.rand(n, device='cuda', generator=gen) < 0.3
  counter = torch.zeros(4, dtype=torch.int32, device='cuda')
  flag = torch.zeros(4, dtype=torch.int32).pin_memory()
  word = flag.numpy().view(np.uint32)
  stream = _lib.raw_stream(is_last.device)
  for seq in (1, 2, 3, 0x7FFFFFF0):
    value = (torch.randn(n, row_elems, device='cuda', generator=gen) * 4).to(dtype)
    out = torch.zeros(n, row_elems, dtype=dtype).pin_memory()
    _lib.fast.emb_mask_actions_notify(value.data_ptr(), out.data_ptr(), n, row_elems, _DTYPE_CODE[dtype],
                                      is_last.data_ptr(), counter.data_ptr(), flag.data_ptr(), seq, stream)
    spins = 0
    while int(word[0]) != seq:                # the host sees the word ...
      spins += 1
      assert spins < 50_000_000, 'the word never arrived'
    want = mask_actions(value, is_last).cpu()   # ... and by then every row (no synchronize in between)
    assert torch.equal(out.view(torch.uint8), want.view(torch.uint8)), (seq, dtype)
    torch.cuda.synchronize()
    assert int(counter[0].item()) == 0        # left zeroed for the next launch
  with pytest.raises(ValueError):
    _lib.fast.emb_mask_actions_notify(value.data_ptr(), out.data_ptr(), n, row_elems, _DTYPE_CODE[dtype],
                                      is_last.data_ptr(), None, flag.data_ptr(), 9, stream)
