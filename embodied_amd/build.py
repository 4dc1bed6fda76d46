"""Build libembodied_hip.so in-tree with hipcc for gfx950.

    python embodied_amd/build.py [--force]

One hipcc invocation per translation unit (parallel), then a link.  The .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import fcntl
import os
import pathlib
import shutil
import subprocess
import sys
import sysconfig

HERE = pathlib.Path(__file__).resolve().parent
CSRC = HERE / 'csrc'
OUT = HERE / 'libembodied_hip.so'
OBJ = HERE / 'build'
SOURCES = ['kernels.hip', 'direct_comm.hip', 'replay_abi.cpp', 'index_abi.cpp', 'kernels_abi.cpp', 'comm_abi.cpp']
ARCH = 'gfx950'
# CPython call shim for the hottest entry points (csrc/fastcall.c): plain C,
# links against nothing; the package falls back to ctypes without it.
FASTCALL = HERE / ('_emb_fastcall' + (sysconfig.get_config_var('EXT_SUFFIX') or '.so'))


# Hot host modules compiled with Cython from their unchanged .py sources
# (_compiled_finder.py): the per-step Python of the vectorised loop.
COMPILED = [
    'embodied_amd.core.driver', 'embodied_amd.core.replay', 'embodied_amd.core.streams',
    'embodied_amd.ops', 'embodied_amd.scans', 'embodied_amd.distributed',
    'embodied_amd.envs.synthetic', 'embodied_amd.core.limiters', 'embodied_amd._lib',
]   # (a compiled module's __file__ points into _compiled/: _lib allows for that)
COMPILED_DIR = HERE / '_compiled'


def hipcc():
  for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if cand and pathlib.Path(cand).exists():
      return cand
  raise RuntimeError('hipcc not found: cannot build libembodied_hip.so')


def scratch_users(remarks):
  """(kernel, bytes per lane) for every kernel the compiler gave scratch memory,
  from -Rpass-analysis=kernel-resource-usage remarks."""
  import re
  out, name = [], None
  for line in remarks.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
      name = m.group(1)
    m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
    if m and int(m.group(1)) > 0:
      out.append((name, int(m.group(1))))
  return out


def stale():
  if not OUT.exists() or not FASTCALL.exists():
    return True
  built = min(OUT.stat().st_mtime, FASTCALL.stat().st_mtime)
  deps = list(CSRC.glob('*')) + [HERE.parent / 'include' / 'embodied_hip.h']
  return any(p.stat().st_mtime > built for p in deps)


def build(force=False, verbose=True):
  if not force and not stale():
    return OUT
  cc = hipcc()
  OBJ.mkdir(exist_ok=True)
  # One builder at a time (pytest-xdist workers, torchrun ranks).
  lock = open(OBJ / '.lock', 'w')
  fcntl.flock(lock, fcntl.LOCK_EX)
  if not force and not stale():
    return OUT
  flags = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-Wall', '-fvisibility=hidden',
           '-fvisibility-inlines-hidden', '-ffunction-sections',
           '-Wno-unused-function', '-Wno-pass-failed', '-x', 'hip']
  # Kernel-argument preload (gfx940+): the first 16 dwords of a kernel's
  # arguments arrive in SGPRs with the wave instead of being fetched by every
  # wave's first s_load.  With host-resident kernel arguments that fetch is a
  # PCIe round trip per wave: preloading takes 1-2 us off every small kernel and
  # off the movers that read their plan through one pointer argument.
  # EMB_KERNARG_PRELOAD=0 builds without it.
  preload = os.environ.get('EMB_KERNARG_PRELOAD', '16')
  if preload not in ('', '0'):
    flags += ['-mllvm', f'-amdgpu-kernarg-preload-count={int(preload)}']

  def compile_one(name):
    obj = OBJ / (name + '.o')
    extra = ['-Rpass-analysis=kernel-resource-usage'] if name.endswith('.hip') else []
    cmd = [cc, *flags, *extra, '-c', str(CSRC / name), '-o', str(obj)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
      raise RuntimeError(f'{" ".join(cmd)}\n{res.stdout}\n{res.stderr}')
    spills = scratch_users(res.stderr)
    if spills:
      # A by-value argument block whose address escapes, or a register spill,
      # turns a 10 us mover into a 250 us one: never ship THAT silently.  Other
      # kernels (scans, env, obs stack) only get a warning -- a small spill on
      # another compiler version must not make the package uninstallable --
      # unless EMB_STRICT_SCRATCH=1 (what this repo's own builds use).
      listing = '\n'.join(f'  {kernel}: {nbytes} bytes/lane' for kernel, nbytes in spills)
      movers = [k for k, _ in spills if any(
          tag in (k or '') for tag in ('flat_move_kernel', 'span_move_kernel',
                                       'obs_stack_insert_kernel'))]
      if movers or os.environ.get('EMB_STRICT_SCRATCH') == '1':
        raise RuntimeError('kernels that use scratch memory:\n' + listing)
      print('warning: kernels that use scratch memory:\n' + listing, file=sys.stderr)
    import re
    rest = [l for l in res.stderr.splitlines() if 'kernel-resource-usage' not in l
            and not re.match(r'^\s*(\d+\s*)?\|', l) and not l.lstrip().startswith('^')
            and 'remarks generated' not in l]
    if verbose and any(l.strip() for l in rest):
      print('\n'.join(rest), file=sys.stderr)
    return obj

  with concurrent.futures.ThreadPoolExecutor(len(SOURCES)) as pool:
    objs = list(pool.map(compile_one, SOURCES))
  tmp = OUT.with_suffix('.so.tmp')
  cmd = [cc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-Wl,--gc-sections', '-s', *map(str, objs), '-o', str(tmp)]
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode:
    raise RuntimeError(f'{" ".join(cmd)}\n{res.stdout}\n{res.stderr}')
  os.replace(tmp, OUT)
  if verbose:
    print(f'built {OUT}')
  build_fastcall(verbose)
  return OUT


def build_fastcall(verbose=True):
  """gcc -shared of csrc/fastcall.c against this interpreter's headers."""
  gcc = shutil.which('gcc') or shutil.which('cc')
  include = sysconfig.get_paths().get('include')
  if not gcc or not include or not (pathlib.Path(include) / 'Python.h').exists():
    if verbose:
      print('no C compiler / Python.h: _emb_fastcall not built (ctypes binding only)', file=sys.stderr)
    FASTCALL.touch()        # an empty file: "tried", import fails, ctypes is used
    return None
  tmp = FASTCALL.with_name(FASTCALL.name + '.tmp')
  cmd = [gcc, '-O2', '-shared', '-fPIC', '-Wall', f'-I{include}', str(CSRC / 'fastcall.c'),
         '-o', str(tmp)]
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode:
    raise RuntimeError(f'{" ".join(cmd)}\n{res.stdout}\n{res.stderr}')
  os.replace(tmp, FASTCALL)
  if verbose:
    print(f'built {FASTCALL}')
  return FASTCALL


def compiled_stale():
  import hashlib
  import json
  try:
    manifest = json.loads((COMPILED_DIR / 'manifest.json').read_text())
  except (OSError, ValueError):
    return True
  for name in COMPILED:
    entry = manifest.get(name)
    source = HERE.parent / (name.replace('.', '/') + '.py')
    if (not entry or not (COMPILED_DIR / entry['file']).exists()
        or hashlib.sha256(source.read_bytes()).hexdigest() != entry['sha256']):
      return True
  return False


def build_compiled(force=False, verbose=True):
  """Cython -> C -> gcc -shared for every module in COMPILED; the manifest with
  the sources' digests is written last.  Any failure leaves the package on its
  plain .py modules (and says so): this is an accelerator, not a dependency."""
  import hashlib
  import json
  if not force and not compiled_stale():
    return COMPILED_DIR
  gcc = shutil.which('gcc') or shutil.which('cc')
  include = sysconfig.get_paths().get('include')
  try:
    from Cython.Compiler import Options
    from Cython.Compiler.Main import compile as cython_compile
  except Exception as e:      # no Cython: plain modules
    if verbose:
      print(f'Cython not importable ({e}): host modules stay plain Python', file=sys.stderr)
    return None
  if not gcc or not include or not (pathlib.Path(include) / 'Python.h').exists():
    if verbose:
      print('no C compiler / Python.h: host modules stay plain Python', file=sys.stderr)
    return None
  COMPILED_DIR.mkdir(exist_ok=True)
  work = OBJ / 'cython'
  work.mkdir(parents=True, exist_ok=True)
  suffix = sysconfig.get_config_var('EXT_SUFFIX') or '.so'

  def to_c(name):
    source = HERE.parent / (name.replace('.', '/') + '.py')
    text = source.read_bytes()
    c_file = work / (name + '.c')
    options = Options.CompilationOptions(
        Options.default_options, output_file=str(c_file), language_level=3,
        compiler_directives={'binding': True, 'language_level': 3})
    result = cython_compile(str(source), options, full_module_name=name)
    if result.num_errors:
      raise RuntimeError(f'cython failed on {source}')
    return name, c_file, hashlib.sha256(text).hexdigest()

  def to_binary(job):
    name, c_file, digest = job
    binary = COMPILED_DIR / (name + suffix)
    tmp = binary.with_name(binary.name + '.tmp')
    cmd = [gcc, '-O2', '-shared', '-fPIC', '-fwrapv', '-w', f'-I{include}', str(c_file), '-o', str(tmp)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
      raise RuntimeError(f'{" ".join(cmd)}\n{res.stdout}\n{res.stderr}')
    os.replace(tmp, binary)
    return name, {'file': binary.name, 'source': name.replace('.', '/') + '.py', 'sha256': digest}

  manifest_path = COMPILED_DIR / 'manifest.json'
  if manifest_path.exists():
    manifest_path.unlink()          # nothing is trusted while binaries are being replaced
  try:
    jobs = [to_c(name) for name in COMPILED]            # Cython's compiler is not thread-safe ...
    with concurrent.futures.ThreadPoolExecutor(min(len(jobs), os.cpu_count() or 2)) as pool:
      manifest = dict(pool.map(to_binary, jobs))        # ... the C compiler runs are independent
  except Exception as e:
    if verbose:
      print(f'compiling the host modules failed, they stay plain Python: {e}', file=sys.stderr)
    return None
  keep = {entry['file'] for entry in manifest.values()}
  for leftover in COMPILED_DIR.glob('*' + suffix):      # modules no longer in the table
    if leftover.name not in keep:
      leftover.unlink()
  manifest_path.write_text(json.dumps(manifest, indent=1))
  if verbose:
    print(f'built {len(manifest)} host modules in {COMPILED_DIR}')
  return COMPILED_DIR


if __name__ == '__main__':
  build(force='--force' in sys.argv)
  build_compiled(force='--force' in sys.argv)
