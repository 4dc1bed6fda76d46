"""The GAE / lambda-return kernels of two builds of the library side by side
(plain ctypes, no package import):  python tools/scan_ab.py libA.so libB.so
(an older build: `git worktree add /tmp/old <commit>; python /tmp/old/embodied_amd/build.py`
and copy its libembodied_hip.so somewhere under the repo so that it travels).
Round 3 used it to find a 10 % regression of the GAE kernel at (65 536, 64) --
a `s_waitcnt vmcnt(0)` between the float and the flag loads after the op struct
changed -- and to confirm the fix (13.2 -> 14.6 -> 13.0 us)."""
import ctypes as C
import sys

import torch

B, T = 65536, 64
rew = torch.randn(B, T, device='cuda')
val = torch.randn(B, T, device='cuda')
flags = torch.rand(B, T, device='cuda') < 0.01
adv = torch.empty(B, T - 1, device='cuda')
tar = torch.empty(B, T - 1, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
p = C.c_void_p
for path in sys.argv[1:]:
  lib = C.CDLL(path)
  for name, call in (
      ('gae', lambda: lib.emb_scan_gae(p(rew.data_ptr()), p(val.data_ptr()), p(flags.data_ptr()), p(flags.data_ptr()),
                                        C.c_int64(B), C.c_int64(T), C.c_float(0.995), C.c_float(0.8),
                                        p(adv.data_ptr()), p(tar.data_ptr()), p(stream))),
      ('lambda', lambda: lib.emb_scan_lambda(p(flags.data_ptr()), p(flags.data_ptr()), p(rew.data_ptr()), p(val.data_ptr()),
                                             C.c_int64(B), C.c_int64(T), C.c_float(0.997), C.c_float(0.95),
                                             p(adv.data_ptr()), p(stream)))):
    for rep in range(3):
      for _ in range(10):
        call()
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      for _ in range(200):
        call()
      b.record()
      torch.cuda.synchronize()
      print(f'{path[-40:]:40s} {name:7s} {a.elapsed_time(b) * 5:.2f} us per launch')
