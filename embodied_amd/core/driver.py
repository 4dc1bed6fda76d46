"""Vectorised env loop with the reference's interface
(embodied/core/driver.py:9-137).

Three ways to run the same loop:

* `Driver(fns, device=None)` — host plumbing exactly like the reference: numpy
  stack, numpy mask, per-env callbacks (BASELINE config "numpy Driver on CPU").
* `Driver(fns, device='cuda')` — per-env observations are written into a pinned
  (N, S) slab (by the env processes themselves with `parallel=True`: one shared,
  HIP-registered block), uploaded -- in pieces, under the slower workers' steps,
  by a kernel that reads the pinned memory -- and handed to the policy as device
  tensors; the action mask (`emb_mask_actions`) runs on the GPU and its stores
  bring the next step's actions down to pinned host memory; the build's own
  `Replay.add` callback is served by ONE batched scatter per step instead of N
  Python calls.
* `Driver(batch_env=env, device='cuda')` — a device-resident vector env
  (`step(acts) -> dict of (N, ...) tensors`): nothing leaves HBM.

Callbacks registered with `on_step(fn)` keep the reference contract
`fn(tran, worker, **kwargs)` per env in order; `on_batch(fn)` receives the
stacked transition once per step.

Lifetime of a step's tensors in device mode.  The reference stacks fresh arrays
every step (driver.py:65), and so does this Driver whenever user code can see
them: with `on_step` / `on_batch` callbacks registered, every transition is made
of tensors that are never written again.  Only when the step's sole consumer is
this package's `Replay.add` (which copies the step into its pool before the next
one) do observations of env processes and masked actions rotate through FOUR
sets of device buffers: a policy that keeps `obs` (or a reader of
`driver.acts`) then sees them overwritten four steps later.
`Driver(..., fresh_obs=True)` switches the rotation off;
a device vector env (`batch_env=`) owns its outputs and states its own rule.
"""
import multiprocessing as mp
import os
from multiprocessing import shared_memory
import time

import cloudpickle
import numpy as np
import torch

from .. import _lib
from .._lib import api, fast
from . import replay as replaylib

# EMB_EARLY_INSERT=0: the Driver does not offer observations to its Replay ahead
# of the policy (every key then goes in with the post-policy insert: the fallback).
# EMB_CARRY_PUBLISH=0: envs that take unmasked actions are served like all others
# (masked copy of the actions, publish launch after the policy: the fallback).
# Both are knobs (`embodied_amd.configure` or the environment), read when the
# first Driver is made.
_EARLY_INSERT = _CARRY = None


def _read_knobs():
  global _EARLY_INSERT, _CARRY
  if _EARLY_INSERT is None:
    _EARLY_INSERT = (_lib.knob('EMB_EARLY_INSERT') or '1') != '0'
    _CARRY = (_lib.knob('EMB_CARRY_PUBLISH') or '1') != '0'

_DTYPE_CODE = {
    torch.uint8: _lib.U8, torch.int8: _lib.I8, torch.int16: _lib.I16,
    torch.int32: _lib.I32, torch.int64: _lib.I64, torch.float16: _lib.F16,
    torch.bfloat16: _lib.BF16, torch.float32: _lib.F32,
    torch.float64: _lib.F64, torch.bool: _lib.BOOL,
}


def cpu_budget():
  """CPUs this process may keep busy: the smaller of its affinity mask and its
  cgroup's CPU quota (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`); a container
  on a 256-CPU host often has 16."""
  try:
    budget = float(len(os.sched_getaffinity(0)))
  except (AttributeError, OSError):
    budget = float(os.cpu_count() or 1)
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      quota, period = f.read().split()[:2]
    if quota != 'max':
      budget = min(budget, int(quota) / int(period))
  except (OSError, ValueError):
    try:
      with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
        quota = int(f.read())
      with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
        period = int(f.read())
      if quota > 0:
        budget = min(budget, quota / period)
    except (OSError, ValueError):
      pass
  return max(1.0, budget)


def auto_envs_per_worker(n_envs):
  """Driver(envs_per_worker='auto'): as many worker processes as the CPU budget
  can run AT ONCE beside the stepping thread (one process per env -- the
  reference's form, driver.py:17-25 -- when the budget allows): more runnable
  processes than CPUs turn a lock-step round trip into time slices."""
  workers = max(1, int(cpu_budget()) - 2)
  return max(1, -(-n_envs // workers))


def mask_actions(value, is_last):
  """value * ~is_last in value's dtype (driver.py:72-74, 84-87)."""
  if torch.is_tensor(value) and value.is_cuda:
    if not value.is_contiguous():
      value = value.contiguous()
    out = torch.empty_like(value)
    n = value.shape[0]
    if value.numel() == 0:
      return out
    fast.emb_mask_actions(
        value.data_ptr(), out.data_ptr(), n, value.numel() // max(n, 1),
        _DTYPE_CODE[value.dtype], is_last.data_ptr(),
        _lib.raw_stream(value.device))
    return out
  if torch.is_tensor(value):
    value = value.numpy()
  keep = ~np.asarray(is_last)
  keep = keep.reshape(keep.shape + (1,) * (value.ndim - keep.ndim))
  return value * keep.astype(value.dtype)


class Driver:

  def __init__(self, make_env_fns=None, parallel=True, device=None,
               batch_env=None, shared_obs=True, fresh_obs=None, envs_per_worker=1,
               upload_groups=None, acts_by_store=None, worker_spin_us=None, acts_notify=True, **kwargs):
    self.kwargs = kwargs
    # (device mode with env processes; both default to the measured optimum: the
    # observation slab goes up in `upload_groups` pieces under the workers' steps,
    # actions come down by kernel stores into pinned memory -- False / 1: one copy each)
    self._upload_groups = _UPLOAD_GROUPS if upload_groups is None else max(1, int(upload_groups))
    self._upload_groups_given = upload_groups
    self._acts_notify = bool(acts_notify)
    self._spin_us = worker_spin_us if worker_spin_us is None else max(0, int(worker_spin_us))
    if acts_by_store is not None and not acts_by_store:
      self._acts_by_store = False
    _read_knobs()
    self._fresh_obs = fresh_obs
    self.device = torch.device(device) if device is not None else None
    if self.device is not None and self.device.type == 'cuda' and self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    self.batch_env = batch_env
    self.parallel = parallel and batch_env is None
    self._upload_src, self._upload_ring, self._upload_turn = None, [], 0
    self._acts_on_host, self._acts_pinned, self._acts_landed = None, {}, None
    self._acts_flag, self._acts_flag_want, self._acts_seq = None, 0, 0
    if batch_env is not None:
      assert self.device is not None and self.device.type == 'cuda'
      self.length = len(batch_env)
      self.act_space = batch_env.act_space
    else:
      assert len(make_env_fns) >= 1
      self.length = len(make_env_fns)
      if self.parallel:
        # envs_per_worker = K > 1 (an addition; the reference starts one process
        # per env, driver.py:17-25): with W = ceil(N / K) worker processes, worker w
        # steps envs w, w + W, w + 2W, ... one after the other -- so rows [0, W) of
        # the observation slab are complete when every worker has stepped its first
        # env, rows [W, 2W) after the second, ...: each such group goes up to the
        # device while the workers step the next one (`_begin_upload`).  For hosts whose CPU budget is smaller than the env count (a
        # container quota): N processes that all become runnable at once then
        # cost more in wake-ups and time slices than their env steps take.
        # Needs the shared-memory step protocol (`shared_obs`).
        if envs_per_worker == 'auto':
          envs_per_worker = auto_envs_per_worker(self.length)
        self._per_worker = K = max(1, int(envs_per_worker))
        if K > 1 and not shared_obs:
          raise ValueError('envs_per_worker > 1 needs shared_obs=True')
        context = mp.get_context()
        n_workers = -(-self.length // K)
        self.pipes, pipes = zip(*[context.Pipe() for _ in range(n_workers)])
        fns = [cloudpickle.dumps(fn) for fn in make_env_fns]
        self._wake = [context.Semaphore(0) for _ in range(n_workers)]
        self.procs = [
            context.Process(target=_env_server,
                            args=(list(range(w, self.length, n_workers)) if K > 1 else w, pipe,
                                  fns[w::n_workers] if K > 1 else fns[w: w + 1], self._wake, w),
                            daemon=True)
            for w, pipe in enumerate(pipes)]
        [proc.start() for proc in self.procs]
        self.pipes[0].send(('act_space',))
        self.act_space = self._receive(self.pipes[0])
        self._shared, self._act_slab, self._fast = {}, {}, False
        if shared_obs:
          self._attach_shared_slab()
        if K > 1 and not self._fast:
          self.close()
          raise RuntimeError('envs_per_worker > 1: the shared-memory step protocol could not be set up '
                             '(the envs must expose obs_space)')
      else:
        self.envs = [fn() for fn in make_env_fns]
        self.act_space = self.envs[0].act_space
    self.callbacks = []
    self.batch_callbacks = []
    self._sinks = []          # per registered callback: its Replay if it is a bare Replay.add
    self.acts = None
    self.carry = None
    self._slab = {}
    self._obs_names, self._obs_has_logs = None, False
    self._uploaded, self._upload_pending = None, False
    self._mask_ring = None
    self._unmasked = None
    self._workers = np.arange(self.length, dtype=np.int64)
    self._workers.setflags(write=False)     # lets Replay.add_batch keep its converted copy
    self.reset()

  # driver.py:34-39
  def reset(self, init_policy=None):
    if self.batch_env is not None:
      self.acts = {
          k: torch.zeros((self.length, *v.shape), dtype=replaylib._TORCH_OF[np.dtype(v.dtype)],
                         device=self.device)
          for k, v in self.act_space.items()}
      self.acts['reset'] = torch.ones(self.length, dtype=torch.bool, device=self.device)
    else:
      self.acts = {
          k: np.zeros((self.length,) + tuple(v.shape), v.dtype)
          for k, v in self.act_space.items()}
      self.acts['reset'] = np.ones(self.length, bool)
    self._acts_on_host = None          # actions fetched ahead belong to the old episode
    self._acts_flag_want = 0
    self.carry = init_policy and init_policy(self.length)

  def _attach_shared_slab(self):
    """Env workers write their observations straight into one shared (N, ...)
    slab per key instead of pickling ~S bytes per env per step through a pipe
    (driver.py:17-25,61-62).  In device mode the slab is registered with the
    HIP runtime, so it is also the pinned source of the upload."""
    self.pipes[0].send(('obs_space',))
    try:
      space = self._receive(self.pipes[0])
      spec = {
          k: (tuple(v.shape), np.dtype(v.dtype).str) for k, v in space.items()
          if not k.startswith('log/')}
    except Exception:
      return
    # ONE block for all observation keys (each key's (N, ...) array at a
    # 256-byte aligned offset): in device mode the whole step's observations go
    # up with a single host-to-device copy instead of one per key.
    layout, offset = {}, 0
    for key, (shape, dtype) in spec.items():
      nbytes = max(1, self.length * int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize)
      layout[key] = [None, shape, dtype, offset, nbytes]
      offset += -(-nbytes // 256) * 256
    total = max(offset, 256)
    block = shared_memory.SharedMemory(create=True, size=total)
    for key, entry in layout.items():
      entry[0] = block.name
      _, shape, dtype, at, _ = entry
      self._shared[key] = (block, np.ndarray((self.length, *shape), dtype, buffer=block.buf, offset=at))
    layout = {k: tuple(v) for k, v in layout.items()}
    if self.device is not None:
      whole = torch.from_numpy(np.ndarray(total, np.uint8, buffer=block.buf))
      if torch.cuda.cudart().cudaHostRegister(whole.data_ptr(), total, 0) == 0:
        self._registered = getattr(self, '_registered', []) + [whole.data_ptr()]
      # Four device copies used in turn (like a vector env's own output ring)
      # while nobody but the Replay sink sees them (`_rotate`); fresh ones otherwise.
      self._upload_layout = layout
      self._upload_src = whole
      self._upload_ring, self._upload_turn = [], 0
      for _ in range(4):
        dev = torch.empty(total, dtype=torch.uint8, device=self.device)
        views = {}
        for key, (_, shape, dtype, at, nbytes) in layout.items():
          kind = replaylib._TORCH_OF[np.dtype(dtype)]
          views[key] = dev[at: at + nbytes].view(kind).view(self.length, *shape)
        self._upload_ring.append((dev, views))
    # Actions go down and completion flags come up through shared memory too:
    # per step the parent copies (N, ...) actions into the action slabs, bumps a
    # sequence number, wakes the workers (one semaphore each) and polls the
    # per-env `done` words — no pickling, no pipe round trip per env.
    acts = {k: (tuple(v.shape), np.dtype(v.dtype).str) for k, v in self.act_space.items()}
    acts['reset'] = ((), np.dtype(bool).str)
    act_layout = {}
    for key, (shape, dtype) in acts.items():
      nbytes = max(1, self.length * int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize)
      block = shared_memory.SharedMemory(create=True, size=nbytes)
      self._act_slab[key] = (block, np.ndarray((self.length, *shape), dtype, buffer=block.buf))
      act_layout[key] = (block.name, shape, dtype)
    n_workers = len(self.pipes)
    if self._spin_us is None:
      # spinning workers must each have a CPU of their own beside the stepping thread:
      # more of them than the budget would spin in the way of workers that still step
      self._spin_us = _SPIN_US if n_workers <= int(cpu_budget()) - 1 else 0
    self._ctrl_block = shared_memory.SharedMemory(create=True, size=8 * (2 + 2 * self.length + n_workers))
    self._ctrl = np.ndarray(2 + 2 * self.length + n_workers, np.int64, buffer=self._ctrl_block.buf)
    self._ctrl[:] = 0
    self._done = self._ctrl[2: 2 + self.length]
    self._extra = self._ctrl[2 + self.length: 2 + 2 * self.length]
    # per worker: 1 while it sleeps on its semaphore (it spins on the sequence word for
    # a while after a step first, see _env_server); ctrl[1] = how long, in microseconds
    self._asleep = self._ctrl[2 + 2 * self.length:]
    self._asleep[:] = 1
    self._ctrl[1] = int(self._spin_us)
    self._seq = 0
    [pipe.send(('attach', layout, self.length, act_layout, self._ctrl_block.name))
     for pipe in self.pipes]
    [self._receive(pipe) for pipe in self.pipes]
    self._fast = True

  def __del__(self):
    try:
      if getattr(self, 'parallel', False) and getattr(self, '_shared', None):
        self.close()
    except Exception:
      pass

  def close(self):
    if self.batch_env is not None:
      getattr(self.batch_env, 'close', lambda: None)()
    elif self.parallel:
      [proc.kill() for proc in self.procs]
      for ptr in getattr(self, '_registered', []):
        torch.cuda.cudart().cudaHostUnregister(ptr)
      self._registered = []
      blocks = []
      for b, _ in list(getattr(self, '_shared', {}).values()) + list(getattr(self, '_act_slab', {}).values()):
        if not any(b is seen for seen in blocks):
          blocks.append(b)
      self._upload_src, self._upload_ring = None, []
      if getattr(self, '_ctrl_block', None) is not None:
        self._ctrl = self._done = self._extra = None
        blocks.append(self._ctrl_block)
        self._ctrl_block = None
      self._shared, self._act_slab = {}, {}
      for block in blocks:
        try:
          block.close()
          block.unlink()
        except Exception:
          pass
    else:
      [env.close() for env in self.envs]

  def on_step(self, callback):
    """fn(tran, worker, **kwargs) per env (driver.py:47-48).  A bound
    `embodied_amd.Replay.add` is recognised and served in batched form."""
    owner = getattr(callback, '__self__', None)
    if (isinstance(owner, replaylib.Replay)
        and getattr(callback, '__func__', None) is replaylib.Replay.add):
      # (host mode too: the stacked host arrays of a step are staged by one call)
      self.batch_callbacks.append(
          lambda trans, workers, **kw: owner.add_batch(trans, workers))
      self._sinks.append(owner)
    else:
      self.callbacks.append(callback)
      self._sinks.append(None)

  def _rotate(self):
    """May this step's tensors come from the rotating buffers?  Only while the
    one consumer of a step is a Replay.add sink (it copies the step into its
    pool before the next one): user callbacks may keep what they are given, as
    they may with the reference's freshly stacked arrays."""
    if self._fresh_obs is not None:
      return not self._fresh_obs
    return len(self._sinks) == 1 and self._sinks[0] is not None and not self.callbacks

  def on_batch(self, callback):
    """fn(trans, workers, **kwargs) once per step with (N, ...) values."""
    self.batch_callbacks.append(callback)
    self._sinks.append(None)

  def __call__(self, policy, steps=0, episodes=0):
    step, episode = 0, 0
    self._count_episodes = episodes > 0
    while step < steps or episode < episodes:
      step, episode = self._step(policy, step, episode)

  # driver.py:55-82
  def _step(self, policy, step, episode):
    if self.batch_env is not None:
      return self._step_device_env(policy, step, episode)
    acts = self.acts
    assert all(len(x) == self.length for x in acts.values())
    host = self._acts_on_host
    if host is not None:
      # The transfers of the last step's (masked) actions into pinned memory were
      # queued right behind the policy; the replay insert's host work ran
      # meanwhile.  Now they are needed.
      self._wait_acts()
      self._acts_on_host = None
    else:
      host = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in acts.items()}
    assert all(isinstance(v, np.ndarray) for v in host.values())
    self._wait_uploads()        # env results overwrite the slab the last upload read
    if self.parallel and self._fast:
      results = self._step_workers(host)
    else:
      per_env = [{k: v[i] for k, v in host.items()} for i in range(self.length)]
      if self.parallel:
        [pipe.send(('step', act)) for pipe, act in zip(self.pipes, per_env)]
        results = [self._receive(pipe) for pipe in self.pipes]
      else:
        results = [env.step(act) for env, act in zip(self.envs, per_env)]
    obs = self._stack(results)
    logs = {k: v for k, v in obs.items() if k.startswith('log/')}
    obs = {k: v for k, v in obs.items() if not k.startswith('log/')}
    assert all(len(x) == self.length for x in obs.values()), obs
    if (self.device is not None and _EARLY_INSERT and len(self._sinks) == 1
        and self._sinks[0] is not None and not logs and not self.callbacks):
      self._sinks[0].offer(obs, self._workers)     # see _step_device_env
    self.carry, acts, outs = policy(self.carry, obs, **self.kwargs)
    assert all(k not in acts for k in outs), (list(outs.keys()), list(acts.keys()))
    is_last = obs['is_last']
    if self.device is not None:
      acts = {k: self._to_device(v) for k, v in acts.items()}
      outs = {k: self._to_device(v) for k, v in outs.items()}
      ended = self._host_flags['is_last']
      if ended.any():
        acts = {k: mask_actions(v, is_last) for k, v in acts.items()}
      # (`reset` may alias the step's is_last, as with a device vector env: device flags
      # are never mutated in place, and with the rotating upload buffers the tensor lives
      # four steps -- user callbacks, which may keep it, get fresh tensors every step)
      self.acts = {**acts, 'reset': is_last if self._rotate() else is_last.clone()}
    else:
      ended = is_last
      if ended.any():
        acts = {k: mask_actions(v, is_last) for k, v in acts.items()}
      self.acts = {**acts, 'reset': is_last.copy()}
    if self.device is not None:
      self._fetch_acts()
      if self._upload_pending == 'unrecorded':
        if self._acts_on_host is not None:
          # the event behind the actions' transfer also covers the uploads queued before
          # it on the same stream; it is waited for before the slab is written again
          self._upload_pending = False
        else:
          self._uploaded.record()
          self._upload_pending = True
    trans = {**obs, **acts, **outs, **logs}
    self._dispatch(trans)
    return step + self.length, episode + int(ended.sum())

  def _wait_acts(self):
    """The next step's actions have landed in pinned memory: the word the last store
    kernel wrote behind its rows says so (`emb_mask_actions_notify`; no event), or
    the event recorded behind the copies does."""
    want = self._acts_flag_want
    if want:
      flag, spins = self._acts_flag_view, 0
      while int(flag[0]) != want:
        spins += 1
        if spins > 200000:          # (~tens of ms without the word: whatever happened, the stream knows)
          torch.cuda.current_stream(self.device).synchronize()
          if int(flag[0]) != want:
            raise RuntimeError('Driver: the kernel that brings the actions down to the host did not report')
          break
      self._acts_flag_want = 0
    else:
      self._acts_landed.synchronize()

  def _fetch_acts(self):
    """Device mode with host envs: start bringing the next step's actions to
    pinned host memory now (asynchronously, behind the policy's kernels), so
    that the transfer and its latency run under the replay insert's host work
    instead of in front of the next env step.

    Each action key travels as ONE small kernel that stores `value * ~is_last`
    straight into the pinned (device-mapped) buffer -- `emb_mask_actions` with a
    host destination -- instead of a device-to-host copy per key through the DMA
    engines (14 us for 256 bytes on an idle GPU, most of it the copy's set-up);
    `reset` is the step's `is_last`, which the host has already.  The first step
    checks the stores against a plain copy and keeps the copies if they differ
    (a pinned allocation the GPU cannot address)."""
    if not all(torch.is_tensor(v) and v.is_cuda for v in self.acts.values()):
      self._acts_on_host = None
      return
    host = {}
    flags = getattr(self, '_host_flags', {}).get('is_last')
    reset = self.acts.get('reset')
    by_store = getattr(self, '_acts_by_store', None)
    stream = None
    stored_all, last_store = True, None
    for k, v in self.acts.items():
      if k == 'reset' and flags is not None:
        host[k] = flags.copy()
        continue
      entry = self._acts_pinned.get(k)
      if entry is None or entry[0].shape != v.shape or entry[0].dtype != v.dtype:
        pinned = torch.empty(v.shape, dtype=v.dtype).pin_memory()
        entry = self._acts_pinned[k] = (pinned, pinned.numpy(), pinned.data_ptr(), _DTYPE_CODE.get(v.dtype))
      pinned, view, ptr, code = entry
      stored = False
      if (by_store is not False and k != 'reset' and reset is not None and code is not None
          and v.is_contiguous() and v.numel() and reset.dtype in (torch.bool, torch.uint8)):
        n = v.shape[0]
        if stream is None:
          stream = _lib.raw_stream(v.device)
        if by_store is None:
          # once: the same values by store and by copy must agree
          try:
            probe = torch.empty(v.shape, dtype=v.dtype).pin_memory()
            fast.emb_mask_actions(v.data_ptr(), probe.data_ptr(), n, v.numel() // n, code, reset.data_ptr(), stream)
            want = mask_actions(v, reset).cpu()
            torch.cuda.current_stream(v.device).synchronize()
            by_store = self._acts_by_store = bool(torch.equal(probe.view(torch.uint8), want.view(torch.uint8)))
          except Exception:
            by_store = self._acts_by_store = False
        if by_store:
          if last_store is not None:
            fast.emb_mask_actions(*last_store)
          last_store = (v.data_ptr(), ptr, n, v.numel() // n, code, reset.data_ptr(), stream)
          stored = True
      if not stored:
        pinned.copy_(v, non_blocking=True)
        stored_all = False
      host[k] = view
    if last_store is not None and stored_all and self._acts_notify:
      # Every key goes down by stores: the LAST launch writes a word behind its rows (the
      # earlier ones finished before it, same stream), the host will read that word --
      # no event record (3.7 us of host time) and no event wait.
      if self._acts_flag is None:
        self._acts_flag = torch.zeros(16, dtype=torch.int32).pin_memory()
        self._acts_flag_view = self._acts_flag.numpy().view(np.uint32)
        self._acts_counter = torch.zeros(16, dtype=torch.int32, device=self.device)
      self._acts_seq = seq = (self._acts_seq % 0x7FFFFFF0) + 1
      a, o, n, e, c, f, st = last_store
      fast.emb_mask_actions_notify(a, o, n, e, c, f, self._acts_counter.data_ptr(), self._acts_flag.data_ptr(), seq, st)
      self._acts_flag_want = seq
      self._acts_on_host = host
      return
    if last_store is not None:
      fast.emb_mask_actions(*last_store)
    if self._acts_landed is None:
      self._acts_landed = torch.cuda.Event()
    self._acts_landed.record()
    self._acts_on_host = host

  def _step_device_env(self, policy, step, episode):
    """Same step with a device-resident vector env: nothing is read back, so
    the mask runs unconditionally (a no-op where nothing ended) and episodes
    are counted only when the caller stops on them."""
    obs = self.batch_env.step(self.acts)
    names = tuple(obs)
    if names != self._obs_names:          # which keys are 'log/*' is looked at once per key set
      self._obs_names = names
      self._obs_has_logs = any(k.startswith('log/') for k in names)
    logs = {}
    if self._obs_has_logs:
      logs = {k: v for k, v in obs.items() if k.startswith('log/')}
      obs = {k: v for k, v in obs.items() if not k.startswith('log/')}
    # The only consumer of the step is one Replay: it is told about the step's
    # observations before the policy runs (an agent that builds its policy
    # batch with ops.obs_stack then writes them to their pool rows in the same
    # launch), and the action mask rides in its insert launch (the pool rows and
    # the next step's actions both receive value * ~is_last) instead of taking a
    # launch of its own.
    sink = self._sinks[0] if len(self._sinks) == 1 and not logs else None
    if sink is not None and _EARLY_INSERT:
      sink.offer(obs, self._workers)
    if self.kwargs:
      self.carry, acts, outs = policy(self.carry, obs, **self.kwargs)
    else:
      self.carry, acts, outs = policy(self.carry, obs)
    if outs:
      assert all(k not in acts for k in outs), (list(outs.keys()), list(acts.keys()))
    is_last = obs['is_last']
    device, Tensor = self.device, torch.Tensor
    for v in acts.values():
      if type(v) is not Tensor or v.device != device:
        acts = {k: self._to_device(v) for k, v in acts.items()}
        break
    if sink is not None:
      # (The masked actions rotate through four sets of buffers, like a vector
      # env's own outputs: a set is overwritten four steps later.  The sink is the
      # step's only consumer; `fresh_obs=True` gives `driver.acts` fresh tensors.)
      if self._unmasked is None:
        # An env that takes the policy's actions as they are, together with
        # `reset` (it ignores the action of an env it resets, as the Env protocol
        # asks: base.py:44-52), needs no masked copy of them: the Replay stores
        # value * ~is_last in its pool rows and may carry that write into its
        # next launch (Replay.carry_publish) -- one launch less per step.
        self._unmasked = bool(
            _CARRY and getattr(self.batch_env, 'takes_unmasked_actions', False)
            and hasattr(sink, 'carry_publish') and hasattr(sink, 'add_step'))
        if self._unmasked:
          sink.carry_publish(True)
      if self._unmasked:
        sink.add_step(obs, acts, outs, self._workers, is_last, False)
        self.acts = {**acts, 'reset': is_last}
        step += self.length
        if self._count_episodes:
          episode += int(is_last.sum().item())
        return step, episode
      names = tuple(acts)
      if self._fresh_obs:
        mask = (names, is_last)
      else:
        ring = self._mask_ring
        if ring is None or ring[0] != names:
          ring = self._mask_ring = (names, [{} for _ in range(4)], [0])
        ring[2][0] = turn = (ring[2][0] + 1) & 3
        mask = (names, is_last, ring[1][turn])
      if len(mask) == 3:
        acts = sink.add_step(obs, acts, outs, self._workers, is_last, mask[2])
      else:
        acts = sink.add_batch({**obs, **acts, **outs}, self._workers, mask=mask)
      self.acts = {**acts, 'reset': is_last}
    else:
      acts = {k: mask_actions(v, is_last) for k, v in acts.items()}
      # Device flags are never mutated in place: `reset` may alias is_last.
      self.acts = {**acts, 'reset': is_last}
      self._dispatch({**obs, **acts, **outs, **logs})
    step += self.length
    if self._count_episodes:
      episode += int(is_last.sum().item())
    return step, episode

  def _dispatch(self, trans):
    for fn in self.batch_callbacks:
      fn(trans, self._workers, **self.kwargs)
    if self.callbacks:
      for i in range(self.length):
        tran = {k: v[i] for k, v in trans.items()}
        [fn(tran, i, **self.kwargs) for fn in self.callbacks]

  def _step_workers(self, host):
    """One step of every env process through shared memory (see
    `_attach_shared_slab`).  Returns the per-env leftovers (keys that are not in
    the observation slab, e.g. 'log/*'), usually empty dicts."""
    for key, (_, slab) in self._act_slab.items():
      slab[...] = host[key]
    self._seq += 1
    seq = self._seq
    self._ctrl[0] = seq
    # Workers that stepped a moment ago are still spinning on the sequence word and
    # are off already (no wake-up latency: a futex wake + the scheduler were ~10 us
    # per level of the tree).  Only when every worker sleeps -- the first step, the
    # step after a pause -- the wake-up goes out as a tree (a semaphore release is
    # ~1.4 us: 64 of them in a row were 90 us of a 250 us step): the Driver wakes
    # the first _FANOUT workers, every worker wakes its own children before it steps
    # its env.  Mixed: each sleeper is woken directly.
    asleep = self._asleep
    if self._spin_us == 0 or asleep.all():
      for wake in self._wake[:_FANOUT]:
        wake.release()
    elif asleep.any():
      for w in np.flatnonzero(asleep):
        self._wake[w].release()
    done, deadline, spins = self._done, None, 0
    pieces = self._begin_upload()
    while True:
      state = done == seq
      # The observation slab goes up in pieces, each as soon as the envs that write
      # it are through (they finish over the ~50 us of the wake tree): the copies run
      # under the slower workers' steps instead of after the last one.
      while pieces and state[pieces[0][0]: pieces[0][1]].all():
        self._upload_piece(pieces.pop(0))
      if state.all():
        break
      if (done < 0).any():
        break
      spins += 1
      if spins > 2000:                 # ~ms: stop burning the core
        time.sleep(0.0002)
        if deadline is None:
          deadline = time.time() + 600
        elif time.time() > deadline:
          [proc.kill() for proc in self.procs]
          raise RuntimeError('env workers did not answer within 600 s')
    results = []
    for i in range(self.length):
      if done[i] < 0 or self._extra[i]:
        results.append(self._receive(self.pipes[i % len(self.pipes)]))    # raises on ('error', e)
      else:
        results.append({})
    return results

  def _upload_piece(self, piece):
    """One piece of the pinned slab to its place in the device buffer.  With more
    than one piece per step: a KERNEL that reads the pinned memory across PCIe
    (emb_copy_bytes) -- a copy through the DMA engines costs ~10 us of set-up per
    call here, five of them per step were slower than one (profiles/
    r06_ab_hostenvs.txt); the single whole-slab copy stays with the DMA engines."""
    _, _, dst, src = piece
    if self._upload_by_kernel:
      fast.emb_copy_bytes(src.data_ptr(), dst.data_ptr(), src.numel(), _lib.raw_stream(self.device))
    else:
      dst.copy_(src, non_blocking=True)

  def _begin_upload(self):
    """Device mode with the shared slab: pick this step's device buffer and return
    the pieces of the slab as (first env, end env, device view, pinned view), in
    env order -- the widest key cut into _UPLOAD_GROUPS row ranges, the rest of
    the block (the narrow keys of every env) last.  [] when there is no slab."""
    self._upload_dev = None
    if self.device is None or self._upload_src is None:
      return []
    plan = getattr(self, '_upload_plan', None)
    if plan is None:
      total, n = self._upload_src.numel(), self.length
      _, _, _, at, nbytes = max(self._upload_layout.values(), key=lambda entry: entry[4])
      groups = self._upload_groups if nbytes >= (256 << 10) and n >= 2 * self._upload_groups else 1
      per = -(-n // groups)
      if (self.parallel and getattr(self, '_per_worker', 1) > 1 and nbytes >= (256 << 10)
          and getattr(self, '_upload_groups_given', None) is None):
        # several envs per worker process: rows [j * W, (j + 1) * W) are the workers' j-th
        # envs and complete together -- one piece each, uploaded under the (j + 1)-th
        per, groups = len(self.pipes), -(-n // len(self.pipes))
      if groups == 1:
        plan = [(0, n, 0, total)]
      else:
        row = nbytes // n
        plan = [(g * per, min(n, (g + 1) * per), at + g * per * row, at + min(n, (g + 1) * per) * row)
                for g in range(groups) if g * per < n]
        plan += [(0, n, a, b) for a, b in ((0, at), (at + nbytes, total)) if b > a]
      self._upload_plan = plan
      self._upload_pieces = {}
      self._upload_by_kernel = len(plan) > 1 and bool(getattr(self, '_registered', None))
    if self._rotate():
      self._upload_turn = turn = (self._upload_turn + 1) & 3
      dev, self._upload_views = self._upload_ring[turn]
      pieces = self._upload_pieces.get(turn)
      if pieces is None:       # the views of this ring slot, made once
        pieces = self._upload_pieces[turn] = [
            (lo, hi, dev[a:b], self._upload_src[a:b]) for lo, hi, a, b in plan]
    else:
      dev = torch.empty(self._upload_src.numel(), dtype=torch.uint8, device=self.device)
      self._upload_views = {
          key: dev[at: at + nbytes].view(replaylib._TORCH_OF[np.dtype(dtype)]).view(self.length, *shape)
          for key, (_, shape, dtype, at, nbytes) in self._upload_layout.items()}
      pieces = [(lo, hi, dev[a:b], self._upload_src[a:b]) for lo, hi, a, b in plan]
    self._upload_dev = dev
    return list(pieces)

  def _to_device(self, value):
    if torch.is_tensor(value):
      if value.device == self.device:
        return value
      return value.to(self.device, non_blocking=True)
    return torch.from_numpy(np.ascontiguousarray(value)).to(self.device, non_blocking=True)

  def _stack(self, results):
    """np.stack of the per-env dicts (driver.py:65).  In device mode rows are
    written straight into a pinned (N, ...) slab per key and uploaded with one
    async copy each."""
    shared = getattr(self, '_shared', None) if self.parallel else None
    keys = list(results[0].keys())
    if self.device is None:
      # With a shared slab the rows are already stacked: one copy per key.
      out = {k: view.copy() for k, (_, view) in shared.items()} if shared else {}
      out.update({k: np.stack([r[k] for r in results]) for k in keys})
      return out
    out, self._host_flags = {}, {}
    if shared:
      for k in ('is_first', 'is_last', 'is_terminal'):
        if k in shared:
          self._host_flags[k] = shared[k][1].copy()
      # The observation slab: uploaded piece by piece while the workers were still
      # stepping (_step_workers / _begin_upload) when the step protocol runs through
      # shared memory; else in one copy here.
      if getattr(self, '_upload_dev', None) is None:
        for piece in self._begin_upload():
          self._upload_piece(piece)
      out.update(self._upload_views)
      self._upload_dev = None
    for k in keys:
      first = np.asarray(results[0][k])
      slab = self._slab.get(k)
      if slab is None or slab[1].shape[1:] != first.shape or slab[1].dtype != first.dtype:
        pinned = torch.empty(
            (self.length, *first.shape), dtype=replaylib._TORCH_OF[first.dtype]).pin_memory()
        slab = self._slab[k] = (pinned, pinned.numpy())
      pinned, view = slab
      for i, r in enumerate(results):
        view[i] = r[k]
      if k in ('is_first', 'is_last', 'is_terminal'):
        self._host_flags[k] = view.copy()
      out[k] = pinned.to(self.device, non_blocking=True)
    # The slab is overwritten by the next step's env results: uploads must be
    # done by then.  One event per step, waited for right before the slab is
    # written again (`_wait_uploads`), so the policy call overlaps the copies; when
    # the step's actions go down to the host behind an event of their own, that one
    # serves (`_step`): an event record is 3.7 us of host time.
    if self._uploaded is None:
      self._uploaded = torch.cuda.Event()
    self._upload_pending = 'unrecorded'
    return out

  def _wait_uploads(self):
    if self._upload_pending == 'unrecorded':      # (nobody recorded behind the uploads: do it now)
      self._uploaded.record()
      self._upload_pending = True
    if self._upload_pending:
      self._uploaded.synchronize()
      self._upload_pending = False

  def _receive(self, pipe):
    """The payload of a worker's ('result', payload) reply.  Anything else — an
    ('error', e) reply, a dead pipe, an unknown tag — ends the run: every
    worker is killed before the exception propagates (driver.py:89-99)."""
    failure = None
    try:
      tag, payload = pipe.recv()
    except Exception as e:                      # EOF / broken pipe: worker is gone
      failure = e
    else:
      if tag == 'result':
        return payload
      failure = RuntimeError(payload) if tag == 'error' else RuntimeError(
          f'env worker sent an unexpected {tag!r} message')
    print('Terminating workers due to an exception.')
    for proc in self.procs:
      proc.kill()
    raise failure


_FANOUT = 8
_SPIN_US = 400          # how long a worker spins on the sequence word after a step before it sleeps
_UPLOAD_GROUPS = 2      # pieces the widest observation key is uploaded in (64 envs x 28 KB: 900 KB each;
                        # 2 measured ahead of 4, 8 and 1: profiles/r06_ab_hostenvs.txt)


def _wake_children(envid, wakes, asleep=None):
  """Workers (envid + 1) * _FANOUT ... + _FANOUT - 1 (heap order under the Driver);
  with `asleep`: those of them that sleep."""
  first = (envid + 1) * _FANOUT
  for child in range(first, min(first + _FANOUT, len(wakes))):
    if asleep is None or asleep[child]:
      wakes[child].release()


def _env_server(envid, pipe, ctor, wakes=None, worker=None):
  """Worker process.  Pipe protocol as the reference's (driver.py:101-137):
  ('step', act) -> ('result', obs), 'obs_space', 'act_space'.  After
  ('attach', obs layout, n, act layout, ctrl name) it switches to the shared
  memory protocol: wait on `wake`, read its action row, step, write the
  observation row, publish `done[envid] = seq`.  `ctor` may be a LIST of pickled
  constructors (Driver(envs_per_worker=K)) with `envid` the list of their slab rows
  (w, w + W, w + 2W, ...): stepped one after the other per wake-up; the pipe protocol
  then speaks for the first."""
  env = None
  envs = []
  ctors = ctor if isinstance(ctor, (list, tuple)) else [ctor]
  rows = list(envid) if isinstance(envid, (list, tuple)) else None      # Driver(envs_per_worker=K): this worker's slab rows
  if rows is not None:
    envid = rows[0]
  worker = envid if worker is None else worker
  blocks, slabs = [], {}

  def open_block(name):
    # Attaching must leave the resource tracker alone: the block is the parent's
    # to unlink.  A tracker of this process's own (spawn) would unlink it at exit
    # and warn about "leaks"; the parent's tracker (fork: shared) holds the name
    # once -- registering here and unregistering again would take the PARENT's
    # entry away, and its unlink would then raise KeyError inside the tracker.
    # (Python 3.13 has SharedMemory(track=False) for this.)
    from multiprocessing import resource_tracker
    register = resource_tracker.register
    resource_tracker.register = lambda *a, **k: None
    try:
      block = shared_memory.SharedMemory(name=name)
    finally:
      resource_tracker.register = register
    blocks.append(block)
    return block

  opened = {}

  def attach(layout, n):
    out = {}
    for key, (name, shape, dtype, *where) in layout.items():
      if name not in opened:
        opened[name] = open_block(name)
      out[key] = np.ndarray((n, *shape), dtype, buffer=opened[name].buf,
                            offset=where[0] if where else 0)
    return out

  def put(obs, row=None):
    row = envid if row is None else row
    rest = {}
    for key, value in obs.items():
      slab = slabs.get(key)
      if slab is None:
        rest[key] = value
      else:
        slab[row] = value
    return rest

  def serve_shared(layout, n, act_layout=None, ctrl_name=None):
    """'attach': from now on observations go into the shared slab; with action
    slabs and a control block the whole step protocol does."""
    slabs.update(attach(layout, n))
    if act_layout is None or wakes is None:
      pipe.send(('result', True))
      return
    acts = attach(act_layout, n)
    act_rows = [(k, v, v.ndim > 1) for k, v in acts.items()]
    ctrl_block = open_block(ctrl_name)
    ctrl = np.ndarray(ctrl_block.size // 8, np.int64, buffer=ctrl_block.buf)
    done, extra = ctrl[2: 2 + n], ctrl[2 + n: 2 + 2 * n]
    asleep = ctrl[2 + 2 * n:]
    spin_s = float(ctrl[1]) * 1e-6
    seen = int(ctrl[0])             # (before the reply: the Driver may start its first step right behind it)
    asleep[worker] = 0
    pipe.send(('result', True))
    wake = wakes[worker]
    clock = time.perf_counter
    while True:
      # After a step: spin on the sequence word for a while -- in a stepping loop the
      # next step arrives within it and starts without a wake-up -- then sleep on the
      # semaphore.  `asleep` tells the Driver whom to wake.  (A wake-up that crosses
      # the announcement is caught by the re-check and, at worst, by the timed wait.)
      slept = False
      until = clock() + spin_s
      while int(ctrl[0]) == seen:
        if clock() < until:
          continue
        slept = True
        asleep[worker] = 1
        if int(ctrl[0]) != seen:
          break
        wake.acquire(timeout=0.05)
        until = 0.0                       # (asleep once: straight back to sleep until a step comes)
      if slept:
        asleep[worker] = 0
        _wake_children(worker, wakes, asleep)       # first: they are woken even if an env then fails
      seq = seen = int(ctrl[0])
      for j, one in enumerate(envs):
        row = rows[j] if rows is not None else envid + j
        try:
          # (a row of a one-dimensional action array is a numpy scalar, a copy already;
          # rows of wider ones are views of the shared slab and are copied)
          obs = one.step({k: (v[row].copy() if wide else v[row]) for k, v, wide in act_rows})
          rest = None
          for key, value in obs.items():
            slab = slabs.get(key)
            if slab is not None:
              slab[row] = value
            elif rest is None:
              rest = {key: value}
            else:
              rest[key] = value
          if rest:
            pipe.send(('result', rest))
            extra[row] = 1
          elif extra[row]:
            extra[row] = 0
          done[row] = seq
        except Exception as e:
          pipe.send(('error', e))
          done[row] = -1
          raise

  def step(action):
    obs = env.step(action)
    pipe.send(('result', put(obs) if slabs else obs))

  handlers = {
      'step': step,
      'attach': serve_shared,
      'obs_space': lambda: pipe.send(('result', env.obs_space)),
      'act_space': lambda: pipe.send(('result', env.act_space)),
  }
  try:
    envs = [cloudpickle.loads(c)() for c in ctors]
    env = envs[0]
    while True:
      if not pipe.poll(0.1):                    # also notices a vanished parent
        continue
      try:
        request = pipe.recv()
      except EOFError:
        return
      handler = handlers.get(request[0])
      if handler is None:
        raise ValueError(f'Invalid message {request[0]}')
      handler(*request[1:])
  except (ConnectionResetError, BrokenPipeError):
    print('Connection to driver lost')
  except Exception as e:
    try:
      pipe.send(('error', e))
    except Exception:
      pass
    raise
  finally:
    for one in envs:
      try:
        one.close()
      except Exception:
        pass
    slabs.clear()
    for block in blocks:
      try:
        block.close()
      except Exception:
        pass
    pipe.close()
