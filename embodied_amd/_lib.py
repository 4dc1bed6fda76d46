"""ctypes binding of libembodied_hip.so (include/embodied_hip.h).

There is no CPU fallback: if the library is missing or fails to load, importing
this module raises.  Build it with `python embodied_amd/build.py`.
"""
import ctypes as C
import os
import pathlib

# torch bundles its own HIP runtime (same SONAME as /opt/rocm's).  Import it
# first so this library binds to the copy torch uses: two HIP runtimes in one
# process do not share devices, streams or allocations.
import torch  # noqa: F401

HERE = pathlib.Path(__file__).resolve().parent
if HERE.name == '_compiled':      # the Cython copy of this module lives one level down
  HERE = HERE.parent
PATH = pathlib.Path(os.environ.get('EMB_LIB_PATH') or HERE / 'libembodied_hip.so')

OK, ERR_INVALID, ERR_HIP, ERR_EMPTY, ERR_POOL_FULL, ERR_NOT_FOUND, ERR_INTERNAL = (
    0, -1, -2, -3, -4, -5, -6)
STEPID_BYTES = 20

U8, I8, I16, I32, I64, F16, BF16, F32, F64, BOOL = range(10)
LAYOUT_SAME, LAYOUT_CHANNELS_FIRST = 0, 1
MODES = {'train': 0, 'report': 1, 'eval': 2}


class EmbError(RuntimeError):

  def __init__(self, code, message):
    super().__init__(f'libembodied_hip: {message} (status {code})')
    self.code = code


class PoolFull(EmbError):
  pass


class ReplayConfig(C.Structure):
  _fields_ = [
      ('length', C.c_int64), ('capacity', C.c_int64), ('chunksize', C.c_int64),
      ('n_slots', C.c_int64), ('online', C.c_int32), ('reserved', C.c_int32),
      ('uid_hi', C.c_uint64), ('owners', C.c_int64),
      ('workers_per_owner', C.c_int64)]


SAMPLE_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p)
SIZE_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p)
INSERT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.POINTER(C.c_uint8), C.c_int32)
REMOVE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64)
PRIORITIZE_FN = C.CFUNCTYPE(
    None, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.c_int64)


class LambdaProblem(C.Structure):
  """emb_lambda_problem_t: one problem of emb_scan_lambda_multi."""
  _fields_ = [
      ('last', C.c_void_p), ('term', C.c_void_p), ('rew', C.c_void_p), ('boot', C.c_void_p),
      ('ret', C.c_void_p), ('B', C.c_int64), ('T', C.c_int64), ('disc', C.c_float), ('lam', C.c_float)]


class ObsSpec(C.Structure):
  """emb_obs_spec_t: how emb_replay_obs_stack_insert lays out the policy batch."""
  _fields_ = [
      ('pixels', C.c_int64), ('channels', C.c_int64), ('layout', C.c_int32),
      ('out_dtype', C.c_int32), ('scale', C.c_float), ('offset', C.c_float)]


class SelectorCallbacks(C.Structure):
  _fields_ = [
      ('user', C.c_void_p), ('sample', SAMPLE_FN), ('size', SIZE_FN),
      ('insert', INSERT_FN), ('remove', REMOVE_FN),
      ('prioritize', PRIORITIZE_FN)]


def _run_builder():
  import importlib.util
  spec = importlib.util.spec_from_file_location('_emb_build', HERE / 'build.py')
  builder = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(builder)
  builder.build(verbose=False)
  return builder


def _load():
  if not PATH.exists():
    # Git-ignored artefact: compile it in-tree on first use (hipcc cross-compiles
    # gfx950 without a GPU).  No hipcc -> no library -> ImportError, never a
    # CPU fallback.
    try:
      _run_builder()
    except Exception as e:
      raise ImportError(
          f'{PATH} is missing and could not be built ({e}). embodied_amd has no '
          'CPU fallback: build the HIP library with `python embodied_amd/build.py` '
          '(needs hipcc, gfx950).') from e
  try:
    return C.CDLL(str(PATH))
  except OSError as e:
    raise ImportError(f'cannot load {PATH}: {e}') from e


lib = _load()

p, i32, i64, u64, f32, f64 = (
    C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double)
pp = C.POINTER(C.c_void_p)

# name -> argtypes; every function returns int32 status unless listed below.
SIGNATURES = {
    'emb_device_count': [p],
    'emb_configure': [C.c_char_p, C.c_char_p],
    'emb_rng_create': [p, i32, pp],
    'emb_rng_integers': [p, i64, i64, p],
    'emb_rng_random': [p, i64, p],
    'emb_rng_choice': [p, p, i32, i64, p],
    'emb_rng_destroy': [p],
    'emb_np_sum': [p, i64, p],
    'emb_tree_create': [i32, u64, pp],
    'emb_tree_insert': [p, i64, f64],
    'emb_tree_remove': [p, i64],
    'emb_tree_update': [p, i64, f64],
    'emb_tree_sample': [p, p],
    'emb_tree_len': [p, p],
    'emb_tree_root_sum': [p, p],
    'emb_tree_shape': [p, i64, p, p, p],
    'emb_tree_destroy': [p],
    'emb_selector_create_fifo': [pp],
    'emb_selector_create_uniform': [u64, pp],
    'emb_selector_create_prioritized': [f64, f64, i32, f64, i32, u64, pp],
    'emb_selector_create_mixture': [p, p, i32, u64, pp],
    'emb_selector_create_recency': [p, i64, i32, i32, i64, u64, pp],
    'emb_selector_create_callback': [p, pp],
    'emb_selector_insert': [p, i64, p, i32],
    'emb_selector_remove': [p, i64],
    'emb_selector_sample': [p, p],
    'emb_selector_len': [p, p],
    'emb_selector_prioritize': [p, p, p, i64],
    'emb_selector_destroy': [p],
    'emb_replay_create': [p, p, u64, pp],
    'emb_replay_destroy': [p],
    'emb_replay_set_keys': [p, i32, p, p, p],
    'emb_replay_grow': [p, i64, p],
    'emb_replay_add_index': [p, i64, p, p, p, p],
    'emb_replay_sample_index': [p, i64, i32, p, p, p],
    'emb_replay_resolve': [p, i64, p, i64, p, p],
    'emb_replay_prioritize': [p, p, p, i64],
    'emb_replay_len': [p, p],
    'emb_replay_sampler_len': [p, p],
    'emb_replay_online_pending': [p, p],
    'emb_replay_free_slots': [p, p],
    'emb_replay_stats': [p, p, i32],
    'emb_replay_add': [p, i64, p, p, p],
    'emb_replay_add_masked': [p, i64, p, p, i32, p, p, p, p, p],
    'emb_replay_obs_stack_insert': [p, i64, p, i32, p, p, p, p, p, p],
    'emb_replay_publish': [p, i64, p, p, i32, p, p, p, p, u64, p],
    'emb_replay_sample': [p, i64, i32, p, p, p, p],
    'emb_replay_sample_grouped': [p, i64, i32, p, i32, i64, p, p, p],
    'emb_replay_sample_heads': [p, i64, i32, p, p, p, p, p],
    'emb_replay_update': [p, i64, i64, p, i32, p, p, p],
    'emb_replay_gather_rows': [p, p, i64, i64, p, p],
    'emb_replay_scatter_rows': [p, p, i64, i32, p, p, p],
    'emb_replay_profile': [p, i32],
    'emb_replay_multistream': [p, i32],
    'emb_replay_profile_read': [p, p, p, i32],
    'emb_replay_profile_report': [p, i32, p, p, i32, p, i32],
    'emb_replay_complete_all': [p],
    'emb_replay_open_chunks': [p, p],
    'emb_replay_reserve_uids': [p, u64],
    'emb_replay_chunks': [p, i64, p, p, p, p, p, p],
    'emb_replay_load_chunk': [p, u64, u64, i64, i64, p],
    'emb_replay_load_items': [p, u64, i64],
    'emb_obs_stack': [p, p, i64, i64, i64, i32, i32, f32, f32, p, p],
    'emb_mask_actions': [p, p, i64, i64, i32, p, p],
    'emb_rows_gather': [p, i64, p, i64, p, p],
    'emb_rows_scatter': [p, i64, p, i64, p, p],
    'emb_window': [p, p, i64, i64, i64, i64, i64, p],
    'emb_window_keys': [i32, p, p, p, i64, i64, i64, i64, p],
    'emb_scan_gae': [p, p, p, p, i64, i64, f32, f32, p, p, p],
    'emb_scan_gae_grouped': [p, p, p, p, i64, i64, f32, f32, p, p, i64, i64, p],
    'emb_scan_lambda': [p, p, p, p, i64, i64, f32, f32, p, p],
    'emb_scan_lambda_multi': [i32, p, p],
    'emb_replay_carry_publish': [p, i32],
    'emb_replay_settle': [p],
    'emb_scan_director': [p, p, p, i64, i64, f32, f32, p, p],
    'emb_abstract_traj': [p, p, i64, i64, i32, p, p, p],
    'emb_synth_env_step': [p, p, p, p, p, i64, i64, i64, i64, p, p, i32, p],
    'emb_comm_unique_id': [p],
    'emb_comm_init': [p, i32, i32, pp],
    'emb_comm_allgather_traj': [p, p, p, i64, p],
    'emb_comm_allreduce_grads': [p, p, i64, i32, p],
    'emb_comm_allreduce_grads_as': [p, p, i64, i32, i32, p],
    'emb_comm_alltoall_slices': [p, p, p, i64, p],
    'emb_comm_allgather_returns': [p, p, p, i64, p],
    'emb_comm_pmean_scalars': [p, p, i64, p],
    'emb_comm_exchange': [p, p, p, p, i64, p, i64, i32, i32],
    'emb_comm_exchange_gather': [p, p, p, p, i64, p, i64, i32, i32],
    'emb_comm_wait': [p, p],
    'emb_comm_destroy': [p],
    'emb_direct_create': [i32, i32, i64, i64, i32, pp],
    'emb_direct_handle': [p, p],
    'emb_direct_connect': [p, p],
    'emb_direct_allreduce': [p, p, i64, i32, i32, p],
    'emb_direct_alltoall': [p, p, p, i64, p],
    'emb_direct_exchange': [p, p, p, p, i64, p, i64, i32, i32],
    'emb_copy_bytes': [p, p, i64, p],
    'emb_mask_actions_notify': [p, p, i64, i64, i32, p, p, p, C.c_uint32, p],
    'emb_stream_create_on_cus': [i32, i32, p],
    'emb_stream_destroy': [p],
    'emb_direct_allgather': [p, p, p, i64, p],
    'emb_direct_set_timeout': [p, i32],
    'emb_direct_exchange_gather': [p, p, p, p, i64, p, i64, i32, i32],
    'emb_direct_wait': [p, p],
    'emb_direct_status': [p, p],
    'emb_direct_destroy': [p],
}

lib.emb_last_error.restype = C.c_char_p
lib.emb_last_error.argtypes = []
lib.emb_abi_version.restype = i32
lib.emb_abi_version.argtypes = []


def check(status):
  if status == OK:
    return
  message = (lib.emb_last_error() or b'').decode('utf8', 'replace')
  if status == ERR_POOL_FULL:
    raise PoolFull(status, message)
  if status == ERR_NOT_FOUND:
    raise KeyError(message)
  if status == ERR_EMPTY:
    raise IndexError(message)
  if status == ERR_INVALID:
    raise ValueError(f'libembodied_hip: {message}')
  raise EmbError(status, message)


class _Api:
  """`api.emb_xxx(...)` calls the C function and raises on a bad status."""

  def __init__(self):
    for name, argtypes in SIGNATURES.items():
      fn = getattr(lib, name)          # AttributeError if the .so is stale
      fn.argtypes = argtypes
      fn.restype = i32
      setattr(self, name, self._wrap(fn))
    self.raw = lib

  @staticmethod
  def _wrap(fn):
    def call(*args):
      status = fn(*args)
      if status:
        check(status)
    call.raw = fn
    return call


api = _Api()


def _load_fastcall():
  """csrc/fastcall.c: the same exported functions called by address with cheap
  argument conversion (ctypes spends 1-2 us on a 10-12 argument call).  None if
  the shim is not built: `fast` then is the ctypes binding."""
  import importlib.machinery
  import importlib.util
  import sysconfig
  path = HERE / ('_emb_fastcall' + (sysconfig.get_config_var('EXT_SUFFIX') or '.so'))
  if os.environ.get('EMB_NO_FASTCALL') == '1':
    return None
  try:
    if not path.exists() and 'EMB_LIB_PATH' not in os.environ:
      _run_builder().build_fastcall(verbose=False)
    if not path.exists() or path.stat().st_size == 0:
      return None
    loader = importlib.machinery.ExtensionFileLoader('_emb_fastcall', str(path))
    spec = importlib.util.spec_from_loader('_emb_fastcall', loader)
    module = importlib.util.module_from_spec(spec)
    loader.exec_module(module)
    return module
  except Exception:
    return None


def _same_values(values, expected):
  """len(values) == len(expected) and every value IS the expected object."""
  if len(values) != len(expected):
    return False
  for value, want in zip(values.values(), expected):
    if value is not want:
      return False
  return True


class _FastApi:
  """`fast.emb_xxx(...)`: hot entry points through the call shim.  Arguments:
  Python ints (addresses, sizes, handles as `.value`), None, floats, ctypes
  arrays.  Same status handling as `api`."""

  SHAPES = {
      'emb_synth_env_step': 'ints', 'emb_mask_actions': 'ints',
      'emb_replay_add': 'ints', 'emb_replay_add_masked': 'ints',
      'emb_replay_obs_stack_insert': 'ints', 'emb_replay_publish': 'ints',
      'emb_replay_sample': 'ints', 'emb_replay_sample_grouped': 'ints', 'emb_replay_update': 'ints',
      'emb_replay_sample_heads': 'ints',
      'emb_replay_gather_rows': 'ints', 'emb_replay_scatter_rows': 'ints',
      'emb_obs_stack': 'obs_stack', 'emb_scan_gae': 'scan', 'emb_scan_lambda': 'scan',
      'emb_scan_gae_grouped': 'scan', 'emb_scan_lambda_multi': 'ints',
      'emb_comm_exchange': 'ints', 'emb_comm_wait': 'ints',
      'emb_direct_exchange': 'ints', 'emb_direct_wait': 'ints',
      'emb_comm_exchange_gather': 'ints', 'emb_direct_exchange_gather': 'ints',
      'emb_copy_bytes': 'ints', 'emb_mask_actions_notify': 'ints',
  }

  def __init__(self, module):
    self.module = module
    # Replay.add_batch's per-key checks in C (None: the Python loop is used).
    self.columns = getattr(module, 'columns', None)
    self.same_values = getattr(module, 'same_values', None) or _same_values
    for name, shape in self.SHAPES.items():
      if module is None:
        setattr(self, name, getattr(api, name))
        continue
      addr = C.cast(getattr(lib, name), C.c_void_p).value
      setattr(self, name, self._wrap(getattr(module, shape), addr))

  @staticmethod
  def _wrap(invoke, addr):
    def call(*args):
      status = invoke(addr, *args)
      if status:
        check(status)
    return call


fast = _FastApi(_load_fastcall())


def raw_stream(device):
  """hipStream_t of torch's current stream on `device`.  Also makes `device`
  the calling thread's current HIP device: the library launches on whatever
  device is current, and helper threads (Prefetch, savers) start on device 0."""
  index = device.index
  if index is None:
    index = torch._C._cuda_getDevice()
  elif torch._C._cuda_getDevice() != index:
    torch.cuda.set_device(index)
  return torch._C._cuda_getCurrentRawStream(index)


_TEMPLATES = {}


def empty(shape, dtype, device):
  """torch.empty(shape, dtype, device) at about half the host cost:
  empty_like of a cached zero-stride one-element template (same contiguous
  result)."""
  key = (tuple(shape), dtype, device)
  template = _TEMPLATES.get(key)
  if template is None:
    if len(_TEMPLATES) > 4096:
      _TEMPLATES.clear()
    template = _TEMPLATES[key] = torch.empty(1, dtype=dtype, device=device).expand(*key[0])
  return torch.empty_like(template)


def ptr(array):
  """Address of a numpy array's buffer (None -> NULL)."""
  return None if array is None else array.ctypes.data


# Knobs that the Python layer reads (the rest belong to the C library): same
# rules as emb_configure -- one name, read once by the first code that needs it,
# `configure` beats the environment and is refused once the knob is in effect.
PY_KNOBS = ('EMB_EARLY_INSERT', 'EMB_CARRY_PUBLISH')
_py_given, _py_read = {}, set()


def knob(name):
  """Value of a Python-level knob (None = not set); marks it as in effect."""
  assert name in PY_KNOBS, name
  _py_read.add(name)
  if name in _py_given:
    return _py_given[name]
  return os.environ.get(name)


def configure(**knobs):
  """`configure(EMB_DEFER_INDEX=0, EMB_GATHER_STORES='plain')`: the tuning knobs
  (INTEGRATION.md lists them) from the host program instead of the environment.
  Each knob is read once by the first call that needs it; setting it later
  raises, as does a name that is not a knob of this package (the library holds
  the list: csrc/knobs.h)."""
  for name, value in knobs.items():
    if name in PY_KNOBS:
      if name in _py_read:
        raise ValueError(f'libembodied_hip: configure: {name} is already in effect (knobs are read '
                         'once: set them before the first Driver is made)')
      if value is None:
        _py_given.pop(name, None)
      else:
        _py_given[name] = str(value)
      continue
    api.emb_configure(name.encode(), None if value is None else str(value).encode())


def device_count():
  n = C.c_int32(0)
  api.emb_device_count(C.byref(n))
  return n.value
