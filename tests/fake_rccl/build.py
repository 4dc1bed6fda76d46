"""Build tests/fake_rccl/libfake_rccl.so (test infrastructure: the loopback
stand-in for librccl that lets emb_comm_* run with several ranks on ONE GPU).
Host code only; hipcc is used because it knows the HIP and RCCL include paths."""
import pathlib
import shutil
import subprocess

HERE = pathlib.Path(__file__).resolve().parent
SOURCE = HERE / 'fake_rccl.cpp'
OUT = HERE / 'libfake_rccl.so'


def build(force=False):
  if not force and OUT.exists() and OUT.stat().st_mtime >= SOURCE.stat().st_mtime:
    return OUT
  cc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  tmp = OUT.with_suffix('.so.tmp')
  cmd = [cc, '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-x', 'hip', '--offload-arch=gfx950',
         str(SOURCE), '-o', str(tmp)]
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode:
    raise RuntimeError(f'{" ".join(cmd)}\n{res.stdout}\n{res.stderr}')
  tmp.replace(OUT)
  return OUT


if __name__ == '__main__':
  print(build(force=True))
