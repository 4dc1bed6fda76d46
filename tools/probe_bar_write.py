"""Can the CPU store straight into device memory (fine-grained allocation through
the PCIe BAR, as the HIP runtime does for device-resident kernel arguments)?
Each probe runs in a child process: a fault must not take the caller down."""
import subprocess
import sys
import textwrap

PROBE = textwrap.dedent('''
    import ctypes as C, sys, time
    import torch
    torch.zeros(1, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    ptr = C.c_void_p()
    flags = int(sys.argv[1])
    rc = hip.hipExtMallocWithFlags(C.byref(ptr), C.c_size_t(1 << 16), C.c_uint(flags))
    print("alloc rc", rc, hex(ptr.value or 0), flush=True)
    src = (C.c_uint8 * 4096)(*[i % 251 for i in range(4096)])
    C.memmove(ptr, src, 4096)                      # CPU store into the allocation
    print("cpu store ok", flush=True)
    out = torch.empty(4096, dtype=torch.uint8, device="cuda")
    rc = hip.hipMemcpy(C.c_void_p(out.data_ptr()), ptr, C.c_size_t(4096), C.c_int(3))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    print("device sees the bytes:", bool((got == [i % 251 for i in range(4096)]).all()), flush=True)
    t0 = time.perf_counter()
    for _ in range(2000):
      C.memmove(ptr, src, 3840)
    print("3840-byte store: %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6), flush=True)
''')

for name, flags in (('hipDeviceMallocFinegrained', 1), ('hipDeviceMallocUncached', 3), ('hipDeviceMallocDefault', 0)):
  res = subprocess.run([sys.executable, '-c', PROBE, str(flags)], capture_output=True, text=True, timeout=120)
  print(f'== {name}: exit {res.returncode}')
  print(res.stdout.strip())
  if res.returncode:
    print(res.stderr.strip()[-300:])
