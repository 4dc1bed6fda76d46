"""Headline benchmark: env steps/s (+ learner train-steps/s) of the Driver ->
Replay -> return-scan hot path, 64 envs x 84x84x4 uint8 obs per GPU.

    python bench.py [--gpus N --steps K --warmup W]

One "step" = one vectorised Driver step of 64 synthetic device envs (env frame
generation, obs stack/transpose into the policy batch, stub policy, action
mask, Replay insert) plus the train steps the reference's `Ratio(train_ratio /
(B*T))` schedules for those 64 env steps (run/train.py:25-26,69-79); a train
step = Replay.sample(B=16, L=65) + Consec window + GAE scan.  The model
forward/backward is outside the path and is not run.

Prints ONE JSON line (rank 0).  `roofline` is for the Replay-sample gather
kernel, timed with HIP events on its own stream inside the timed region;
`cpu_baseline` is the numpy oracle of the same workload on this box's host.
"""
import argparse
import json
import os
import sys
import time

# Kernel arguments in host memory instead of device memory (a HIP runtime
# setting, read when HIP starts): every launch is ~1.6 us cheaper for the host
# (4.6 -> 3.0 us on MI355X), and the PPO step is launch-bound.  Waves then read
# their arguments across PCIe, which is why the big movers get a device copy
# of their argument block (abi.cpp run_move).  HIP_FORCE_DEV_KERNARG=1/0 in the
# environment overrides; DESIGN.md 4 has both sets of numbers.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '0')


def _flag_value(name, default):
  """`--name value` / `--name=value` from sys.argv (needed before argparse runs:
  the pinning happens ahead of the first import of torch)."""
  for i, arg in enumerate(sys.argv):
    if arg == name and i + 1 < len(sys.argv):
      return sys.argv[i + 1]
    if arg.startswith(name + '='):
      return arg[len(name) + 1:]
  return default


def pin_cpus(width=4):
  """Keep the stepping thread (and the runtime's helper threads, which inherit
  the mask) on a few adjacent, currently idle CPUs.  The GPU boxes are shared
  hosts (256 CPUs, load average 25-40 from other tenants): left to the
  scheduler the same loop measured 2.3-3.0 M env steps/s run to run, pinned
  2.96-3.15 M.  Picks the `width` adjacent CPUs of the allowed set that were
  least busy over 100 ms of /proc/stat, one window per local rank (rank r takes
  the r-th least busy window that does not overlap a better one).
  `--pin 0` leaves the affinity alone, `--pin a-b` sets it."""
  knob = _flag_value('--pin', 'auto')
  if knob == '0' or not hasattr(os, 'sched_setaffinity'):
    return None
  try:
    allowed = sorted(os.sched_getaffinity(0))
    if knob != 'auto':
      lo, _, hi = knob.partition('-')
      chosen = [c for c in range(int(lo), int(hi or lo) + 1) if c in allowed]
    else:
      if len(allowed) < 4 * width:
        return None

      def busy():
        out = {}
        with open('/proc/stat') as f:
          for line in f:
            if line.startswith('cpu') and line[3].isdigit():
              parts = line.split()
              vals = [int(x) for x in parts[1:9]]
              out[int(parts[0][3:])] = (sum(vals) - vals[3] - vals[4], sum(vals))
        return out
      a = busy()
      time.sleep(0.1)
      b = busy()
      load = {c: (b[c][0] - a[c][0]) / max(1, b[c][1] - a[c][1]) for c in allowed if c in a and c in b}
      windows = []
      for i in range(0, len(allowed) - width + 1, width):
        cpus = allowed[i:i + width]
        if cpus[-1] - cpus[0] == width - 1 and all(c in load for c in cpus):
          # (coarse: idle windows tie and fall back to their order, so that ranks
          # measuring at slightly different moments still rank them alike)
          windows.append((round(sum(load[c] for c in cpus), 1), i, cpus))
      if not windows:
        return None
      windows.sort()
      rank = int(os.environ.get('LOCAL_RANK', '0'))
      chosen = windows[min(rank, len(windows) - 1)][2]
    if chosen:
      os.sched_setaffinity(0, chosen)
      return chosen
  except Exception:
    pass
  return None


def _is_launcher():
  """`python bench.py --gpus N` (N > 1) without a launcher re-executes itself
  under torch.distributed.run: the ranks pin themselves, not this process."""
  if 'WORLD_SIZE' in os.environ:
    return False
  gpus = _flag_value('--gpus', '1')
  return gpus.isdigit() and int(gpus) > 1


# (Env worker processes would inherit the mask: the host-env workloads stay unpinned.)
PINNED = (pin_cpus() if __name__ == '__main__' and not _is_launcher()
          and not {'--parallel-envs', '--host-envs'} & set(sys.argv) else None)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # same guide: what a plain device-to-device copy achieves (read + written bytes)


def parse():
  p = argparse.ArgumentParser()
  p.add_argument('--gpus', type=int, default=1)
  p.add_argument('--steps', type=int, default=50000)    # ~2 s timed at ~40 us per step
  p.add_argument('--warmup', type=int, default=2000)
  p.add_argument('--envs', type=int, default=64)
  p.add_argument('--batch', type=int, default=16)
  p.add_argument('--length', type=int, default=64)
  p.add_argument('--context', type=int, default=1)
  p.add_argument('--capacity', type=int, default=100_000)   # ppo/configs.yaml:39
  p.add_argument('--train-ratio', type=float, default=3.0)  # ppo/configs.yaml:51
  # The shipped PPO model at the benchmark's shapes (84x84x4 image, 6 actions), counted module by
  # module from ppo/configs.yaml:86-96 + ppo/agent.py:128-151 + ppo/nets.py:11-80 by tools/count_params.py
  # (profiles/r06_param_counts.txt): impala encoder 4 354 464 + action embedding 8 192 + GRU 7 872 512
  # + policy 6 150 + value 1 025.  (DreamerV3's default size, configs[2]/[3]: 181.7 M / 166.0 M.)
  p.add_argument('--grad-numel', type=int, default=12_242_343)
  p.add_argument('--exchange', default='dp_slice',
                 choices=['dp_slice', 'online', 'trajectories', 'returns', 'none'],
                 help='N>1: how trajectories cross xGMI.  dp_slice (default) = SURVEY 8e\'s cheaper '
                      'form: batches holding fresh on-policy windows are cut into one block per rank '
                      'and exchanged with ONE all-to-all, every rank trains on a slice mixed from '
                      'all ranks\' envs ((n-1)/n * B*L*S bytes each way per exchanged batch).  '
                      'online = all-gather of those batches ((n-1) * B*L*S received); trajectories '
                      '= all-gather of every batch; returns = all-gather of the GAE outputs only')
  p.add_argument('--comm', default='auto', choices=['auto', 'native', 'c10d', 'direct'],
                 help='N>1: who carries the collectives of the timed path.  c10d = torch.distributed '
                      '(ProcessGroupNCCL).  native = the library\'s own RCCL entry points '
                      '(emb_comm_exchange: one call per train step, ~1/3 of the host time).  direct = '
                      'the library\'s direct xGMI schedule (emb_direct_exchange: every rank writes all '
                      'n-1 peers at once through hipIpc pointers, one link each; no RCCL).  auto '
                      '(default): both are checked against torch.distributed on the job\'s own GPUs '
                      'and timed on its own bytes before the timed regions (`native_comm`); the timed '
                      'path takes the faster of those that passed on every rank (direct only if it '
                      'wins by more than 10 %%), c10d if neither did')
  p.add_argument('--grad-dtype', default='f32', choices=['bf16', 'f32'],
                 help='N>1: dtype of the flat gradient buffer that is all-reduced every train '
                      'step.  f32 (default) is the reference\'s precision: it averages f32 gradient '
                      'leaves, embodied/jax/opt.py:52-54.  bf16 halves the bytes on the links and is a '
                      'LOWER precision than the reference: a line taken with it says so in '
                      '`config.parallelism` and is context, not the headline')
  p.add_argument('--cpu-seconds', type=float, default=15.0)
  p.add_argument('--sustained-seconds', type=float, default=10.0,
                 help='after the headline region: the same loop for this long (SURVEY 8d asks for '
                      'wall-clock rates over >= 10 s windows); reported as `sustained`, 0 = skip')
  p.add_argument('--stamp-every', type=int, default=0,
                 help='dispatch stamps on one sample gather in this many (1 = every gather).  A '
                      'stamped launch is not free: its start stamp puts ~10 us of idle time in front '
                      'of the kernel on the GPU\'s timeline (rocprofv3 --kernel-trace of this script: '
                      'marker, gap, gather) and costs the host a few microseconds: with one in 4 stamped '
                      'the dreamer workload ran at 610 k env steps/s, one in 16: 651 k, one in 64: 661 k, '
                      'none: 665 k (ppo: 4.40 / 4.52 / 4.55 / 4.56 M; the gather times themselves do '
                      'not move).  0 (default): 32 when the timed region holds >= 8192 gathers, 16 from '
                      '4096, 4 from 256, every second one in shorter regions')
  p.add_argument('--no-context', action='store_true',
                 help='skip the batches-per-launch sweep and the plain-copy reference after the '
                      'timed regions (counter passes: only the workload\'s own launches)')
  p.add_argument('--prewarm-train-steps', type=int, default=200,
                 help='train steps run after the fill and before --warmup, so that a short '
                      '--warmup still times a warm train path')
  p.add_argument('--no-cpu-baseline', action='store_true')
  p.add_argument('--no-dreamer-leg', action='store_true',
                 help='skip the short configs[2] run (1M-step uniform replay, latents written back) '
                      'that the default single-GPU PPO run appends as `workloads.dreamer`')
  p.add_argument('--dreamer-leg-steps', type=int, default=4000)
  p.add_argument('--prefetch', type=int, default=1, help='train batches gathered per launch')
  p.add_argument('--reuse-outputs', type=int, default=0,
                 help='Replay(reuse_outputs=K): sampled batches rotate through K output sets '
                      '(0 = fresh tensors per sample, as the reference returns fresh arrays)')
  p.add_argument('--consec', type=int, default=1,
                 help='consec_train (ppo/configs.yaml:13): windows served per sampled sequence; '
                      'sampling and windowing are one gather launch')
  p.add_argument('--workload', default='ppo', choices=['ppo', 'dreamer'],
                 help='ppo = BASELINE configs[1] (default); dreamer = configs[2]: 1M-step '
                      'uniform replay, train_ratio 32, 40 KB/step latents written back')
  p.add_argument('--selector', default='uniform', choices=['uniform', 'prioritized'],
                 help='prioritized = ppo/configs.yaml:42 (exponent .8, maxfrac .5, initial inf, '
                      'zero_on_sample) instead of the default fracs.uniform 1.0')
  p.add_argument('--streams', type=int, default=0, choices=[0, 1, 2],
                 help='2: the learner\'s work of a train step -- sample gather, return scans, '
                      'write-back -- is issued on a HIP stream of its own (the actor / learner split '
                      'of run/actor_learner.py; the reference dispatches its train step '
                      'asynchronously as well); 1: everything on one stream; 0 (default): 2 for the '
                      'dreamer workload (+6-7 %%: the env steps\' small kernels run in the ramps of the '
                      '144 MB gathers), 1 for ppo (measured 12 %% SLOWER with 2: the persistent gather '
                      'holds every CU\'s registers for its 11 us, the three dependent small kernels '
                      'of the env step wait behind it instead of beside it)')
  p.add_argument('--learner-cus', type=int, default=-1,
                 help='--streams 2: compute units the learner\'s stream may use (emb_stream_create_on_cus: a '
                      'CU-masked HIP stream; the sample gather sizes its persistent grid by it) so that it '
                      'runs BESIDE the env step\'s small kernels instead of in front of them.  0 = no mask '
                      '(a plain second stream); -1 (default) = the measured optimum of the workload')
  p.add_argument('--learner-first-cu', type=int, default=0, help='with --learner-cus: the first unit of the range')
  p.add_argument('--unmasked-env-actions', action='store_true',
                 help='the synthetic env declares that it takes the policy\'s actions unmasked together with '
                      '`reset` (it ignores the action of an env it resets, Env protocol base.py:44-52): the '
                      'Driver then skips the masked copy and the Replay carries its masked pool write into '
                      'the next launch -- two dependent launches per step instead of three.  NOT the '
                      'reference\'s Driver form (driver.py:72-75 masks before the env sees the action), so '
                      'not the default: `value` is measured on the masked form, this rate rides beside it '
                      'as context (`config.env_actions`)')
  p.add_argument('--mask-actions-for-env', action='store_true',
                 help='(the default since round 6; kept so that older command lines still parse) the env '
                      'receives masked actions, the reference\'s form (driver.py:72-75)')
  p.add_argument('--context-only', action='store_true',
                 help='dreamer workload: Replay(heads={"dyn/": context}) -- the replay-context latents come '
                      'back (B, K, ...) instead of (B, L, ...), which is all the shipped agent reads of them '
                      '(dreamerv3/agent.py:322-331, lhs = x[:, :K]); the write-back still covers every step')
  p.add_argument('--regions', type=int, default=0,
                 help='how many fenced regions of exactly --steps steps are timed back to back; `value` is '
                      'their median.  0 (default) = 16 when --steps < 256 (a 20-step region lasts 0.3 ms and '
                      'reads +-30 %% from run to run by itself), else 1')
  p.add_argument('--pin', default='auto', help='CPUs of the stepping thread: auto (default), 0 = leave alone, a-b')
  p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                 help='N>1: process-group backend (gloo: the control flow on fewer GPUs than ranks)')
  p.add_argument('--force-dist', action='store_true', help='run the RCCL code path with one rank')
  p.add_argument('--no-timer', action='store_true',
                 help='no dispatch stamps (roofline is then null): for a rocprofv3 run that sees the gather '
                      'without the timed-dispatch perturbation')
  p.add_argument('--host-envs', action='store_true',
                 help='step 64 numpy envs on the host and upload through the pinned slab '
                      '(PCIe-inclusive rate; never the headline value)')
  p.add_argument('--parallel-envs', action='store_true',
                 help='with --host-envs: one process per env writing into the shared slab')
  p.add_argument('--envs-per-worker', default='auto',
                 help='with --parallel-envs: this many envs per worker process (Driver(envs_per_worker=K)); '
                      'auto (default) = as many worker processes as the CPU budget (affinity mask, cgroup '
                      'quota) runs at once, one per env if it allows (the reference\'s form, driver.py:17-25)')
  return p.parse_args()


class Ratio:
  """elements.when.Ratio as run/train.py:25-26 uses it [SURVEY App. A]."""

  def __init__(self, ratio):
    self.ratio = ratio
    self.prev = None

  def __call__(self, step):
    if self.ratio <= 0:
      return 0
    if self.prev is None:
      self.prev = step
      return 1
    repeats = int((step - self.prev) * self.ratio)
    self.prev += repeats / self.ratio
    return repeats


def build_path(args, rank, device):
  import embodied_amd as emb
  from embodied_amd.envs import synthetic
  L = args.consec * args.length + args.context
  dreamer = args.workload == 'dreamer'
  selector = None
  if args.selector == 'prioritized':
    selector = emb.selectors.Prioritized(
        exponent=0.8, maxfrac=0.5, initial=float('inf'), zero_on_sample=True, seed=0)
  replay = emb.Replay(
      length=L, capacity=args.capacity, chunksize=1024, online=not dreamer, seed=0,
      selector=selector, device=device, replica=rank, reuse_outputs=args.reuse_outputs,
      heads={'dyn/': args.context} if (dreamer and args.context_only) else None)
  n = args.envs
  if args.host_envs:
    fns = [(lambda e=e: synthetic.HostSyntheticEnv(rank * n + e)) for e in range(n)]
    driver = emb.Driver(fns, parallel=args.parallel_envs, device=device,
                        **({'upload_groups': int(os.environ['EMB_BENCH_UPLOAD_GROUPS'])}
                           if 'EMB_BENCH_UPLOAD_GROUPS' in os.environ else {}),
                        **({'acts_by_store': os.environ['EMB_BENCH_ACTS_BY_STORE'] != '0'}
                           if 'EMB_BENCH_ACTS_BY_STORE' in os.environ else {}),
                        **({'acts_notify': os.environ['EMB_BENCH_ACTS_NOTIFY'] != '0'}
                           if 'EMB_BENCH_ACTS_NOTIFY' in os.environ and args.parallel_envs else {}),
                        **({'worker_spin_us': int(os.environ['EMB_BENCH_WORKER_SPIN_US'])}
                           if 'EMB_BENCH_WORKER_SPIN_US' in os.environ and args.parallel_envs else {}),
                        **({'envs_per_worker': (args.envs_per_worker if args.envs_per_worker == 'auto'
                                                else int(args.envs_per_worker))} if args.parallel_envs else {}))
    env = None
  else:
    # The env owns a 4-deep ring of output buffers (transitions are copied into
    # the replay pool within the step, `reset` aliases the previous is_last).
    env = synthetic.SyntheticBatchEnv(
        n, shape=(84, 84, 4), episode_len=1000, env0=rank * n, device=device, ring=4,
        takes_unmasked_actions=args.unmasked_env_actions)
    driver = emb.Driver(batch_env=env, device=device)
  driver.on_step(replay.add)
  # 4096 pre-drawn action rows (the stub policy hands them out in turn).
  actions = list(torch.randint(0, 6, (4096, n), dtype=torch.int32, device=device).unbind(0))
  state = {'tick': 0}
  if dreamer:   # replay-context latents every policy step (dreamerv3/rssm.py:40-43)
    deter = torch.zeros((n, 8192), dtype=torch.float32, device=device)
    stoch = torch.zeros((n, 32, 64), dtype=torch.float32, device=device)

  # The agent owns its input staging: two policy-batch buffers used in turn.
  staging = [torch.empty((n, 4, 84, 84), dtype=torch.bfloat16, device=device) for _ in range(2)]

  def policy(carry, obs, **kw):
    # obs stack/transpose into the policy batch (N, C, H, W) bf16 in [0, 1]:
    # what the agent does first with the frames (jax/agent.py:230).
    batch = emb.ops.obs_stack(
        obs['image'], layout='channels_first', dtype=torch.bfloat16, scale=1 / 255,
        out=staging[state['tick'] & 1])
    state['tick'] += 1
    state['policy_batch'] = batch
    outs = {'dyn/deter': deter, 'dyn/stoch': stoch} if dreamer else {}
    return carry, {'action': actions[state['tick'] % 4096]}, outs

  return emb, env, replay, driver, policy


def train_steps_local(counters, base, headline):
  return headline['train_steps'] - base['train_steps']


def launch_ranks(args):
  """`python bench.py --gpus N` without a launcher: start N local ranks, one per
  GPU, under torch.distributed.run and become that process (rank 0 prints the
  JSON line).  Reference shape: embodied/jax/internal.py:96-105."""
  import socket
  backend = args.backend
  have = torch.cuda.device_count()
  if backend == 'nccl' and have < args.gpus:
    raise SystemExit(f'bench.py --gpus {args.gpus}: this node has {have} GPU(s); RCCL needs one '
                     'GPU per rank (--backend gloo runs the control flow on fewer)')
  with socket.socket() as sock:
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1', '--master-port', str(port),
         os.path.abspath(__file__), *sys.argv[1:]]
  sys.stdout.flush()
  os.execv(sys.executable, cmd)


def main():
  args = parse()
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    launch_ranks(args)
  if args.streams == 0:
    args.streams = 2 if args.workload == 'dreamer' else 1
  if args.workload == 'dreamer':      # dreamerv3/configs.yaml:11,15,40-42 (size overridden to 1e6)
    if args.capacity == 100_000:
      args.capacity = 1_000_000
    if args.train_ratio == 3.0:
      args.train_ratio = 32.0
    if args.grad_numel == 12_242_343:     # DreamerV3's default size on configs[2]'s inputs (tools/count_params.py)
      args.grad_numel = 181_738_790
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  # (LOCAL_RANK wraps so that the multi-rank control flow can be exercised on a
  # box with fewer GPUs: --backend gloo, several ranks on one device.)
  if rank != 0:
    # stdout carries exactly one JSON line (rank 0's); whatever other ranks or
    # their libraries print (RCCL's NCCL_DEBUG=VERSION banner) goes to stderr.
    os.dup2(sys.stderr.fileno(), sys.stdout.fileno())
  local %= max(torch.cuda.device_count(), 1)
  torch.cuda.set_device(local)
  device = torch.device('cuda', local)
  if args.learner_cus < 0:
    args.learner_cus = 0
  if args.streams == 2 and args.learner_cus > 0:
    # hipExtStreamCreateWithCUMask makes a BLOCKING stream (no flags argument): it and the
    # NULL stream wait for each other at every launch (measured: 52 us per step instead of
    # 13.6 with the Driver on torch's default = NULL stream).  The Driver's side of such a
    # run therefore gets a non-blocking stream of its own, current from here on.
    torch.cuda.set_stream(torch.cuda.Stream(device))
  use_dist = world > 1 or args.force_dist
  if use_dist:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29517')
    import datetime
    backend = args.backend
    dist.init_process_group(
        backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300),
        **({'device_id': device} if backend == 'nccl' else {}))
  assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
  rccl_ranks = dist.get_world_size() if use_dist else 1

  emb, env, replay, driver, policy = build_path(args, rank, device)
  driver_ref = [driver if args.parallel_envs else None]     # (read for the line's config only)
  B, T, L = args.batch, args.length, args.consec * args.length + args.context
  # The learner lends nothing out: a batch is consumed inside its train step, so
  # the stream takes every batch back one draw later (two output sets circulate
  # instead of seven fresh tensors per sample), and the GAE results go into two
  # agent-owned pairs used in turn -- explicit reuse, no liveness guessing.
  stream = iter(emb.streams.Consec(
      emb.streams.Stateless(replay.sample, B * args.prefetch, 'train', recycle=1),
      length=T, consec=args.consec, prefix=args.context, strict=True, contiguous=True))
  lambda_out = [[torch.empty(B * args.prefetch, T + args.context - 1, device=device),
                 torch.empty(B * T, 15, device=device)] for _ in range(2)]
  gae_out = [tuple(torch.empty(B * args.prefetch, T + args.context - 1, device=device) for _ in range(2))
             for _ in range(2)]
  should_train = Ratio(args.train_ratio / (B * T))
  value = torch.randn(B * args.prefetch, T + args.context, device=device)
  imag_rew = torch.randn(B * T, 16, device=device)
  imag_flags = torch.zeros(B * T, 16, dtype=torch.bool, device=device)
  entries = None
  if args.workload == 'dreamer':      # two sets of model outputs used in turn (agent-owned)
    entries = [(torch.zeros((B * args.prefetch, L, 8192), device=device),
                torch.zeros((B * args.prefetch, L, 32, 64), device=device)) for _ in range(2)]
  grad_dtype = torch.bfloat16 if args.grad_dtype == 'bf16' else torch.float32
  grads = (torch.zeros(args.grad_numel, dtype=grad_dtype, device=device)
           if use_dist and args.grad_numel else None)
  counters = {'env_steps': 0, 'train_steps': 0}
  marks = []
  comm = None
  native, native_stuck, native_comm = None, False, None
  if use_dist:
    from embodied_amd import distributed as D
  use_native = use_direct = False
  direct_comm = None
  link = None               # who carries a train step's collectives (set after the self-check)
  collectives = {'on': True, 'sliced': 0, 'gathered': 0}
  schedule = {'fences': 0}          # fences at which the ranks compared their exchange schedules (and agreed)
  fresh = {'on': False, 'stream': None}

  def train_step():
    if args.workload == 'dreamer':
      # sample -> lambda-returns (replay (B,T) and imagination (B*K,H+1)) ->
      # write the new latents back over the sampled steps (agent.py:144-150).
      # With ranks (configs[3]): every rank's Replay is its own, the one
      # collective per train step is the gradient all-reduce, waited for one
      # train step later.
      batch = next(stream)
      # (both return scans of the train step -- replay (B,T) and imagination
      # (B*K,H+1), agent.py:401-405,464-466 -- in one launch)
      adv, _ = emb.scans.lambda_returns(
          [(batch['is_last'], batch['is_terminal'], batch['reward'], None, value, 1 - 1 / 333, 0.95),
           (imag_flags, imag_flags, imag_rew, None, imag_rew, 1 - 1 / 333, 0.95)],
          out=lambda_out[counters['train_steps'] & 1])
      # The train step's new latents (`entries`, agent.py:139-150): outputs of the
      # model, i.e. the agent's own tensors -- not the sampled batch, of whose
      # latents the agent has read [:, :K] only.
      new = entries[counters['train_steps'] & 1]
      replay.update({'stepid': batch['stepid'], 'dyn/deter': new[0], 'dyn/stoch': new[1]})
      if use_dist and grads is not None and collectives['on']:
        link.wait()
        link.exchange(grads=grads)
    elif fresh['on']:
      # (context leg `fresh_batches`: the learner as the reference writes it --
      # every batch and every scan result a fresh allocation, ppo/main.py:262-272)
      batch = next(fresh['stream'])
      adv, tar = emb.scans.gae(
          batch['reward'], value, batch['is_last'], batch['is_terminal'], hor=200, lam=0.8)
    elif not use_dist or not collectives['on']:
      batch = next(stream)
      adv, tar = emb.scans.gae(
          batch['reward'], value, batch['is_last'], batch['is_terminal'], hor=200, lam=0.8,
          out=gae_out[(counters['train_steps'] // args.prefetch) & 1])
    else:
      # Sample straight into one packed buffer so the trajectory exchange is a
      # single RCCL all-gather; both collectives run async on RCCL's stream and
      # are waited for one train step later (the reference returns train outs
      # one step late too: embodied/jax/agent.py:286-294).
      rows = B * args.prefetch
      sliced = (args.exchange == 'dp_slice' and rows % world == 0
                and replay.online_pending() > 0)
      # (packed batches, received slices and scan results rotate through four
      # buffers each: a buffer is read again one train step later at the latest)
      if sliced:
        # Fresh on-policy windows: cut the batch into one block per rank.
        collectives['sliced'] += 1
        flat, batch, layout = D.sample_packed(replay, rows, groups=world, reuse=4)
      else:
        flat, batch, layout = D.sample_packed(replay, rows, reuse=4)
      scan_out = packed_out[counters['train_steps'] // args.prefetch & 3]
      # Last train step's collectives (they ran behind the env steps since):
      # `link` is the library's own RCCL exchange (emb_comm_exchange, its own
      # stream) or the same contract on the process group (D.GroupComm).
      link.wait()
      send = None
      if sliced:
        # This rank's slice of the global batch = block `rank` of every rank's
        # batch; it is complete one train step later (the wait above) and is
        # what the learner then computes returns on.  Slices and gradients go
        # out in ONE exchange call.
        if not received_ring:
          received_ring.extend(torch.empty_like(flat) for _ in range(4))
        received = received_ring[collectives['sliced'] & 3]
        link.exchange(flat, received, grads)
        state_keep[:] = [flat, received]
        late, slice_state['recv'] = slice_state.get('recv'), (received, layout)
        source, source_info = late if late is not None else (flat, layout)
        adv, tar = D.gae_packed(source, source_info, value, hor=200, lam=0.8, out=scan_out)
      else:
        adv, tar = D.gae_packed(flat, layout, value, hor=200, lam=0.8, out=scan_out)
        if args.exchange == 'trajectories' or (args.exchange == 'online' and layout.online.any()):
          send = flat
        elif args.exchange == 'returns':
          returns = returns_ring[counters['train_steps'] // args.prefetch & 3]
          returns[0].copy_(adv)
          returns[1].copy_(tar)
          send = returns.view(torch.uint8).reshape(-1)
        # The all-gather forms (north_star's "all-gather of trajectories"; the reference
        # assembles every process's slice into the global batch, jax/internal.py:145-152)
        # go through the SAME transport as the gradients, in ONE exchange call: RCCL's
        # all-gather on the library's stream (emb_comm_exchange_gather), the direct
        # schedule's push to all n-1 peers (emb_direct_exchange_gather), or the process
        # group (GroupComm) -- whichever carries the timed path.
        if send is not None:
          ring = gathered_ring.setdefault(send.numel(), [])
          slot = collectives['gathered'] & 3
          if slot >= len(ring):
            ring.append(torch.empty(world * send.numel(), dtype=torch.uint8, device=device))
          gathered = ring[slot]
          link.exchange(send, gathered, grads, gather=True)
          collectives['gathered'] += 1
          state_keep[:] = [gathered, send]
        elif grads is not None:
          link.exchange(grads=grads)
      # When the links are the bottleneck the host could queue train steps far
      # ahead of the GPU (one gathered buffer each): stay within 8 train steps.
      if counters['train_steps'] % 4 == 0:       # one mark per 4 train steps, 8 marks deep
        if len(marks) == 8:
          mark = marks.pop(0)
          mark.synchronize()                     # the GPU has passed the mark of 32 train steps ago
        else:
          mark = torch.cuda.Event()
        # (eight events, recorded again in turn; the stream is looked up once:
        # torch.cuda.current_stream() costs more than the record itself)
        if 'stream' not in slice_state:
          slice_state['stream'] = torch.cuda.current_stream(device)
        mark.record(slice_state['stream'])
        marks.append(mark)
    counters['train_steps'] += args.prefetch
    return adv

  state_keep = []
  slice_state = {}
  received_ring = []
  gathered_ring = {}
  returns_ring = [torch.empty(2, B * args.prefetch, T + args.context - 1, device=device)
                  for _ in range(4)] if use_dist else None
  packed_out = [tuple(torch.empty(B * args.prefetch, T + args.context - 1, device=device) for _ in range(2))
                for _ in range(4)] if use_dist else None

  # The learner's stream (--streams 2): the Replay orders its pool accesses across
  # the two streams itself (emb_replay_multistream; abi.cpp StreamOrder).
  # (a low-priority learner stream measured level for PPO)
  learner_cus = max(0, args.learner_cus) if args.streams == 2 else 0
  cu_stream = None
  if learner_cus > 0:
    # The learner confined to a share of the chip (a CU-masked HIP stream): its 10 us sample
    # gather then runs beside the env step's latency-bound kernels instead of in front of them.
    cu_stream = emb.ops.CuStream(learner_cus, first_cu=args.learner_first_cu, device=device)
    learner = cu_stream.stream
  else:
    learner = torch.cuda.Stream(device) if args.streams == 2 else None
  if learner is not None:
    main_stream = torch.cuda.current_stream(device)
    set_stream = torch._C._cuda_setStream
    to_learner = dict(stream_id=learner.stream_id, device_index=learner.device_index,
                      device_type=learner.device_type)
    to_main = dict(stream_id=main_stream.stream_id, device_index=main_stream.device_index,
                   device_type=main_stream.device_type)
    learner.wait_stream(main_stream)       # the buffers made above are ready before the learner uses them

  def one_step():
    driver(policy, steps=args.envs)            # exactly one vectorised step
    counters['env_steps'] += args.envs
    if len(replay) >= B * T:
      repeats = should_train(counters['env_steps'])
      if repeats:
        if learner is None:
          for _ in range(repeats):
            train_step()
        else:
          set_stream(**to_learner)
          try:
            for _ in range(repeats):
              train_step()
          finally:
            set_stream(**to_main)

  # Fill the buffer so sampled windows come from all over HBM, not from cache.
  driver.reset()
  fill = -(-(args.capacity + L) // args.envs) + L
  for _ in range(fill):
    driver(policy, steps=args.envs)
  counters['env_steps'] = fill * args.envs
  # (EMB_RCCL_LIB: the library binds its RCCL symbols from that file -- with the
  # suite's loopback transport the native path also runs between gloo ranks that
  # share one GPU, tests/test_gpu_bench_launcher.py)
  if use_dist and args.comm != 'c10d' and (dist.get_backend() == 'nccl' or os.environ.get('EMB_RCCL_LIB')):
    packed = D.PackedLayout([(k.name, k.dtype, k.shape) for k in replay._keys], B * args.prefetch, L).nbytes
    native, native_stuck, native_comm, direct_comm = native_comm_check(
        rank, world, device, args.grad_numel, grad_dtype,
        B * args.prefetch * L * sum(k.rowbytes for k in replay._keys) // world,
        # the unit of the all-gather forms: one rank's packed batch, or its GAE results
        2 * B * args.prefetch * (T + args.context - 1) * 4 if args.exchange == 'returns' else packed)
    # Which transport carries the timed path.  An explicitly named one that did not
    # pass its check ends the job; `auto` NEVER does: it degrades -- direct, else
    # RCCL through the library, else torch.distributed -- and says why in
    # `native_comm.auto`.  Every input of the choice is the same on every rank (the
    # check's verdicts are MIN-reduced, its times MAX-reduced below).
    if args.comm == 'native' and native_comm is None:
      raise SystemExit(f'--comm native: the self-check did not pass: {native}')
    if args.comm == 'direct' and direct_comm is None:
      raise SystemExit(f'--comm direct: the self-check did not pass: {native.get("direct")}')
    gathering = args.exchange in ('online', 'trajectories', 'returns')
    step_key = 'exchange_gather_step' if gathering else 'exchange_step'
    use_direct = args.comm == 'direct'
    if args.comm == 'auto':
      why = []
      if native_stuck:
        why.append('the self-check never returned (watchdog): torch.distributed')
      if native_comm is None and not native_stuck:
        why.append(f'emb_comm_* did not pass its check ({native.get("status")}'
                   + (f': {native["error"]}' if native.get('error') else '') + ')')
      if direct_comm is None and not native_stuck:
        d = native.get('direct') or {}
        why.append(f'emb_direct_* did not pass its check ({d.get("status", "not run")}'
                   + (f': {d["error"]}' if d.get('error') else '') + ')')
      auto = {}
      if native_comm is not None and direct_comm is not None and world > 1:
        # Both passed against torch.distributed on this job's own GPUs and were timed on
        # the job's own bytes: the one whose train-step exchange (trajectories + gradient
        # all-reduce + wait) is faster on the slowest rank -- direct only if it wins by > 10 %.
        mine = torch.tensor([native['per_call'][f'native_{step_key}']['total_us'],
                             native['direct']['per_call'][f'direct_{step_key}']['total_us']],
                            dtype=torch.float64, device=device)
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        rccl_us, direct_us = mine.tolist()
        use_direct = direct_us < 0.9 * rccl_us
        auto.update({'rccl_exchange_us': round(rccl_us, 1), 'direct_exchange_us': round(direct_us, 1)})
        why.append('both passed; ' + ('direct wins by more than 10 %' if use_direct else 'direct does not win by 10 %'))
      elif direct_comm is not None and native_comm is None:
        use_direct = True
      elif native_comm is not None and direct_comm is not None:
        why.append('one rank: nothing to choose')
      auto['chose'] = 'direct' if use_direct else 'native' if native_comm is not None else 'c10d'
      auto['why'] = '; '.join(why)
      native['auto'] = auto
    use_native = not use_direct and native_comm is not None
    native['timed_path'] = 'direct' if use_direct else 'native' if use_native else 'c10d'
    if native_stuck:
      native['stuck'] = True
  if use_dist:
    link = direct_comm if use_direct else native_comm if use_native else D.GroupComm()
  # --no-timer: no dispatch stamps (roofline is then null).  Stamps are
  # switched on before the warm-up so that the stamps' events exist by then.
  # One gather in --stamp-every carries stamps, unless the timed region is too
  # short for that to leave a usable sample (the driver's --steps 20 has three
  # or four gathers: every second one then).
  expected = args.steps * args.envs * args.train_ratio / (B * T)
  if args.consec != 1:
    stamp_every = 1
  elif args.stamp_every > 0:
    stamp_every = args.stamp_every
  elif expected >= 256:
    stamp_every = 32 if expected >= 8192 else 16 if expected >= 4096 else 4
  elif args.sustained_seconds > 0 and expected >= 2:
    # A handful of gathers per region (the driver's --steps 20 has four, sixty
    # over its sixteen regions).  A stamped launch costs the stepping thread 7.4 us
    # instead of 1.6 (two event records), and a 260 us region has nowhere to hide
    # that: with every second gather stamped `value` read 6.5 % lower than with
    # no stamps at all, with one in eight 1-2 % (profiles/r05_ab_stamps.txt).  The
    # roofline figure of such a run stands on the sustained window's thousands of
    # stamped launches anyway (`roofline.source`); the regions keep one in eight.
    stamp_every = 8
  else:
    # No sustained window behind the regions: every second gather is stamped,
    # beginning with the second (the stamp counter restarts right before the timed
    # region; the region's first gather runs on a GPU the fence has just drained).
    stamp_every = 2 if expected >= 2 else 1
  replay.profile(not args.no_timer, every=stamp_every)
  # Like `timeit`: no pass of the interpreter's cyclic collector over its whole
  # heap (tens of ms: the GPU idles, the first steps after it measure 130 + 100 us
  # instead of 17 + 34) between the warm-up and the end of a timed region that
  # may be 300 us long.  A full collection runs HERE, before the warm-up; the
  # collector is switched on again behind the region's closing fence and stays
  # on for the sustained window (whose rate shows what it costs: nothing).
  import gc
  gc.collect()
  gc.disable()
  # The fill runs no train step: warm the train path (allocator, online queue,
  # caches) whatever --warmup says, then the caller's warmup steps.
  if learner is not None:
    set_stream(**to_learner)
  for _ in range(args.prewarm_train_steps):
    train_step()
  if learner is not None:
    set_stream(**to_main)
  for _ in range(args.warmup):
    one_step()
  replay.profile_read(reset=True)
  if not args.no_timer:
    replay.profile(True, every=stamp_every)       # restart the stamp counter (first stamped gather: the stamp_every-th)
  base = dict(counters)

  def fence():
    if link is not None:
      link.wait()
    torch.cuda.synchronize(device)
    if use_dist:
      # The sliced / not-sliced choice of a train step is taken from local replay
      # state (`online_pending`); the ranks run the same schedule, so they take
      # it alike -- checked here, where the ranks meet anyway: a rank that ever
      # chose differently would have paired its collectives wrongly.
      digest = torch.tensor([float(collectives['sliced']), -float(collectives['sliced']),
                             float(counters['train_steps']), -float(counters['train_steps']),
                             float(collectives['gathered']), -float(collectives['gathered'])],
                            dtype=torch.float64, device=device)
      dist.all_reduce(digest, op=dist.ReduceOp.MAX)      # doubles as the barrier
      hi_s, neg_lo_s, hi_t, neg_lo_t, hi_g, neg_lo_g = digest.tolist()
      if hi_s != -neg_lo_s or hi_t != -neg_lo_t or hi_g != -neg_lo_g:
        raise SystemExit(f'rank {rank}: ranks disagree on the exchange schedule '
                         f'(sliced {-neg_lo_s:.0f}..{hi_s:.0f}, gathered {-neg_lo_g:.0f}..{hi_g:.0f}, '
                         f'train steps {-neg_lo_t:.0f}..{hi_t:.0f})')
      schedule['fences'] += 1
      torch.cuda.synchronize(device)      # (the barrier's own kernels)

  # The timed region: EXACTLY --steps steps between two fences (barrier +
  # synchronize), the maximum over ranks.  A region of a few hundred microseconds
  # (the driver's --steps 20) is dominated by the first launch after a fence and
  # by where its handful of train steps fall: it is then timed --regions times
  # back to back and `value` is the MEDIAN region (all of them in `regions`).
  n_regions = args.regions if args.regions > 0 else (16 if args.steps < 256 else 1)
  fence()
  region_times, region_counts = [], []
  for _ in range(n_regions):
    before = dict(counters)
    start = time.perf_counter()
    for _ in range(args.steps):
      one_step()
    fence()
    took = time.perf_counter() - start
    if use_dist:
      t = torch.tensor([took], dtype=torch.float64, device=device)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      took = float(t.item())
    region_times.append(took)
    region_counts.append({k: counters[k] - before[k] for k in counters})
  gc.enable()
  order = sorted(range(n_regions), key=lambda i: region_times[i])
  median_at = order[(n_regions - 1) // 2]         # (the lower median: a region that was measured)
  elapsed = region_times[median_at]
  total_elapsed = sum(region_times)

  launches, gather_ms, gather_kernel = replay.profile_report('sample', reset=True)
  h_deferred, h_predicted, _ = replay.profile_report('deferred', reset=True)
  h_inline, h_carried, _ = replay.profile_report('carried', reset=True)
  wb_launches, wb_ms, wb_kernel = replay.profile_report('update', reset=True)
  headline = dict(counters)
  was_unmasked = bool(getattr(driver, '_unmasked', False))     # how the timed regions handed actions to the env

  # The same loop for >= --sustained-seconds more (the headline region is as
  # long as --steps says; this window is long whatever the caller chose).
  sustained = None
  if args.sustained_seconds > 0:
    if stamp_every < 16 and args.consec == 1 and not args.no_timer:
      # A short headline region stamps every second gather; the long window that
      # follows holds thousands: one in 16 (a stamp costs the job time, see --stamp-every)
      replay.profile(True, every=16)
      sustained_stamp_every = 16
    else:
      sustained_stamp_every = stamp_every
    fence()
    s_start = time.perf_counter()
    s_steps = 0
    while True:
      for _ in range(256):
        one_step()
      s_steps += 256
      go_on = time.perf_counter() - s_start < args.sustained_seconds
      if use_dist:          # every rank must leave the loop after the same step
        link.wait()         # the process group's collective queues up behind the exchange's
        flag = torch.tensor([1.0 if go_on else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        go_on = bool(flag.item())
      if not go_on:
        break
    s_issued = time.perf_counter()
    fence()
    s_elapsed = time.perf_counter() - s_start
    if use_dist:
      t = torch.tensor([s_elapsed], dtype=torch.float64, device=device)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      s_elapsed = float(t.item())
    s_launches, s_ms, _ = replay.profile_report('sample', reset=True)
    s_deferred, s_predicted, _ = replay.profile_report('deferred', reset=True)
    s_inline, s_carried, _ = replay.profile_report('carried', reset=True)
    s_wb_launches, s_wb_ms, _ = replay.profile_report('update', reset=True)
    sustained = {
        'seconds': round(s_elapsed, 3), 'steps': s_steps,
        'env_steps_per_s': round(s_steps * args.envs * world / s_elapsed, 1),
        'train_steps_per_s': round(
            (counters['train_steps'] - headline['train_steps']) * world / s_elapsed, 2),
        'ms_per_step': round(s_elapsed / s_steps * 1e3, 5),
        'gather_avg_us': round(s_ms / s_launches * 1e3, 2) if s_launches else None,
        'gather_launches': s_launches, 'stamped_one_in': sustained_stamp_every,
        # of the window's publishes: bookkeeping on the helper thread / next early insert on
        # predicted rows / action write carried into the next launch (all) / (rode along)
        'publishes': {'deferred': s_deferred, 'predicted': int(s_predicted), 'carried': int(s_carried),
                      'carried_inline': s_inline},
        **({'writeback_avg_us': round(s_wb_ms / s_wb_launches * 1e3, 2)} if s_wb_launches else {}),
        # how far the GPU was behind the host when the last step had been issued:
        # a few steps' worth = the host sets the pace, milliseconds = the GPU does
        'closing_fence_us': round((s_start + s_elapsed - s_issued) * 1e6, 1),
    }
  # Context (single GPU, device envs, no part of `value`): the same loop for two more
  # seconds with the OTHER form of the env's input.  `value` is measured on the
  # reference's form -- the env receives masked actions (driver.py:72-75): a masked
  # copy + the Replay's publish launch after the policy, three dependent launches per
  # step.  The other form: an env that declares `takes_unmasked_actions` (it ignores the
  # action of an env it resets, base.py:44-52) gets the policy's actions as they are
  # and the masked pool write rides in the next early-insert launch -- two launches.
  # The stored transitions are the same either way.
  other_env_actions = None
  if (not use_dist and env is not None and args.sustained_seconds > 0 and not args.no_context
      and hasattr(env, 'takes_unmasked_actions') and driver._unmasked is not None):
    fence()
    flipped = not was_unmasked
    env.takes_unmasked_actions = flipped
    driver._unmasked = None
    if not flipped:
      replay.carry_publish(False)
    for _ in range(64):
      one_step()
    fence()
    m_start, m_steps = time.perf_counter(), 0
    while time.perf_counter() - m_start < 2.0:
      for _ in range(256):
        one_step()
      m_steps += 256
    fence()
    m_elapsed = time.perf_counter() - m_start
    assert bool(driver._unmasked) == flipped
    other_env_actions = {
        'form': 'unmasked + reset' if flipped else 'masked',
        'env_steps_per_s': round(m_steps * args.envs / m_elapsed, 1), 'steps': m_steps,
        'ms_per_step': round(m_elapsed / m_steps * 1e3, 5),
        'what': ('same loop, the env takes the policy\'s actions unmasked together with `reset` (Env protocol '
                 'base.py:44-52): no masked copy, the masked pool write carried into the next launch, two '
                 'dependent launches per step' if flipped else
                 'same loop, the env handed masked actions (the reference\'s form): a masked copy and a publish '
                 'launch after the policy, three dependent launches per step')}
    env.takes_unmasked_actions = was_unmasked
    driver._unmasked = None
    if not was_unmasked:
      replay.carry_publish(False)
    for _ in range(16):
      one_step()
    fence()
    replay.profile_read(reset=True)
  # Context (single GPU, ppo, no part of `value`): the same loop for two more
  # seconds with the learner written as the shipped mains write it -- the stream
  # built from `bind(replay.sample, batch, mode)` without `recycle`, GAE without
  # `out=`: seven fresh batch tensors and two fresh results per train step, as
  # the reference's fresh arrays.
  fresh_batches = None
  if (not use_dist and args.workload == 'ppo' and args.sustained_seconds > 0 and not args.no_context
      and args.prefetch == 1):
    import functools
    fence()
    fresh['stream'] = iter(emb.streams.Consec(
        emb.streams.Stateless(functools.partial(replay.sample, B, 'train')),
        length=T, consec=args.consec, prefix=args.context, strict=True, contiguous=True))
    fresh['on'] = True
    for _ in range(64):
      one_step()
    fence()
    f_start, f_steps, f_trains = time.perf_counter(), 0, counters['train_steps']
    while time.perf_counter() - f_start < 2.0:
      for _ in range(256):
        one_step()
      f_steps += 256
    fence()
    f_elapsed = time.perf_counter() - f_start
    fresh['on'] = False
    fresh_batches = {
        'env_steps_per_s': round(f_steps * args.envs / f_elapsed, 1), 'steps': f_steps,
        'train_steps_per_s': round((counters['train_steps'] - f_trains) / f_elapsed, 2),
        'ms_per_step': round(f_elapsed / f_steps * 1e3, 5),
        'what': 'same loop, the learner as ppo/main.py:262-272 spells it: Stateless(bind(replay.sample, '
                'batch, mode)) without recycle, GAE without out= (fresh tensors per train step)'}
  # Ranks only, context: the same loop with the collectives switched off (N
  # independent replicas: no exchange, no gradient all-reduce) -- what the path
  # itself does on N GPUs, next to what the links allow with them.
  replicas_only = None
  sliced_share = min(1.0, collectives['sliced'] / max(1, counters['train_steps']))
  gather_share = min(1.0, collectives['gathered'] / max(1, counters['train_steps']))
  if use_dist and args.sustained_seconds > 0:
    fence()
    collectives['on'] = False
    r_steps = max(256, int(2.0 / max(elapsed / args.steps, 1e-6)) // 256 * 256)   # same on every rank
    r_before = counters['train_steps']
    r_start = time.perf_counter()
    for _ in range(r_steps):
      one_step()
    fence()
    r_elapsed = time.perf_counter() - r_start
    t = torch.tensor([r_elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    r_elapsed = float(t.item())
    collectives['on'] = True
    replay.profile_read(reset=True)
    replicas_only = {
        'seconds': round(r_elapsed, 3), 'steps': r_steps,
        'env_steps_per_s': round(r_steps * args.envs * world / r_elapsed, 1),
        'train_steps_per_s': round((counters['train_steps'] - r_before) * world / r_elapsed, 2),
        'what': 'same loop, collectives off: N independent replicas',
    }
  if native is not None and 'per_call' in native and train_steps_local(counters, base, headline) > 0:
    # Every transport's own time for one train step's collectives at this world size
    # (measured by the self-check on the job's own bytes) against the period at which
    # the timed region issued train steps.
    done = train_steps_local(counters, base, headline)
    traj, share = (('all_gather', gather_share) if args.exchange in ('online', 'trajectories', 'returns')
                   else ('all_to_all', sliced_share))
    per_step = {}
    for name, prefix, calls in (('c10d', 'c10d', native['per_call']), ('rccl', 'native', native['per_call']),
                                ('direct', 'direct', (native.get('direct') or {}).get('per_call') or {})):
      if f'{prefix}_all_reduce' in calls and f'{prefix}_{traj}' in calls:
        per_step[name] = round(calls[f'{prefix}_all_reduce']['total_us'] + share * calls[f'{prefix}_{traj}']['total_us'], 1)
    native['per_train_step'] = {
        **({'collectives_us': per_step['rccl']} if 'rccl' in per_step else {}),
        # the same train step's collectives on the direct schedule (all n-1 links at once) / the process group
        **({'direct_collectives_us': per_step['direct']} if 'direct' in per_step else {}),
        **({'c10d_collectives_us': per_step['c10d']} if 'c10d' in per_step else {}),
        'issue_period_us': round(total_elapsed / done * 1e6, 1),
        'trajectory_collective': traj, 'trajectory_share': round(share, 3), 'sliced_share': round(sliced_share, 3),
    }
    # links bound the job when the transport of the timed path needs longer for one train
    # step's collectives than the path takes to issue the next train step
    on_path = per_step.get({'native': 'rccl'}.get(native.get('timed_path'), native.get('timed_path')))
    if on_path is not None:
      native['per_train_step']['link_bound'] = bool(on_path > native['per_train_step']['issue_period_us'])
  expected = None
  if use_dist and replicas_only is not None and replicas_only['train_steps_per_s'] > 0:
    S_all = sum(k.rowbytes for k in replay._keys)
    expected = scaling_expectation(
        world, (grads.numel() * grads.element_size()) if grads is not None else 0,
        B * args.prefetch * L * S_all,
        sliced_share if args.exchange == 'dp_slice' and args.workload == 'ppo' else 0.0,
        1e6 * world / replicas_only['train_steps_per_s'],
        gather_share if args.workload == 'ppo' else 0.0,
        (2 * B * args.prefetch * (T + args.context - 1) * 4) if args.exchange == 'returns' else None)
    # The same bound with the collectives' MEASURED time per train step instead of a
    # share of link peak, one column per transport: rccl (emb_comm_*) and the direct
    # schedule (emb_direct_*: all n-1 links at once).  x = n * min(1, period / collectives).
    step = (native or {}).get('per_train_step') or {}
    period = expected['train_period_us_collectives_off']
    columns = (('rccl', 'collectives_us'), ('direct', 'direct_collectives_us'), ('c10d', 'c10d_collectives_us'))
    expected['link_bound_x_measured'] = {
        name: round(world * min(1.0, period / max(step[key], 1e-9)), 2) for name, key in columns if key in step}
    expected['collectives_us_measured'] = {name: step[key] for name, key in columns if key in step}
    # the same two readings, measured: speed-up over ONE rank of the replicas_only loop
    per_rank = replicas_only['env_steps_per_s'] / world
    expected['measured_x'] = {
        'value': round(args.steps * args.envs * world / elapsed / per_rank, 2),
        **({'sustained': round(sustained['env_steps_per_s'] / per_rank, 2)} if sustained else {}),
        'replicas_only': float(world),
        'unit': 'x one rank of this run with the collectives off'}
  step_name = 'exchange_gather_step' if args.exchange in ('online', 'trajectories', 'returns') else 'exchange_step'
  counters.update(headline)
  env_steps = args.steps * args.envs * world                     # of ONE region (the median one is `value`)
  train_steps = (counters['train_steps'] - base['train_steps']) * world     # of all regions together
  S = sum(k.rowbytes for k in replay._keys)
  # a key of `Replay(heads=)` moves key_len steps per sequence, the others L
  seq_bytes = sum(k.rowbytes * (replay._key_lens[i] if replay._key_lens is not None else L)
                  for i, k in enumerate(replay._keys))
  # read B*L*S + write B*L*S per sampled batch; with consec > 1 the windows are
  # gathered directly (prefix rows read once per window) and the 1-byte flag
  # keys take a second, tiny launch: normalise per sample, not per launch.
  algo_bytes = 2 * B * args.prefetch * (args.consec * (T + args.context) * S if args.consec != 1 else seq_bytes)
  samples = max(1, (counters['train_steps'] - base['train_steps']) // (args.prefetch * args.consec))
  roofline = None
  traffic, traffic_source = pmc_traffic(algo_bytes)
  if launches:
    per_sample = args.consec != 1      # consec > 1: two launches per sample, normalised per sample
    avg_s = gather_ms / (samples if per_sample else launches) / 1e3
    region = {'avg_launch_us': round(avg_s * 1e6, 2), 'launches': launches,
              'frac': round(algo_bytes / avg_s / 1e9 / HBM_PEAK_GBS, 4)}
    source, one_in = f'stamped launches of the timed region ({args.steps} steps)', stamp_every
    # A short region (the driver's --steps 20 holds three or four gathers, sixty
    # over its sixteen regions, a handful of them stamped) gives a sample that
    # moves by 10 % from run to run:
    # the figure the line stands on is then the mean over the sustained window's
    # stamped launches (same loop, same kernel, >= 1000 of them), which is what
    # `rocprofv3 --kernel-trace --stats` of the same command averages over; the
    # region's own sample stays beside it as `headline_region`.
    if (launches < 256 and not per_sample and sustained is not None
        and (sustained.get('gather_launches') or 0) >= 1000):
      avg_s = sustained['gather_avg_us'] * 1e-6
      launches = sustained['gather_launches']
      one_in = sustained['stamped_one_in']
      region['stamped_one_in'] = stamp_every
      source = f'stamped launches of the sustained window ({sustained["seconds"]} s)'
    achieved = algo_bytes / avg_s / 1e9
    roofline = {
        'bound': 'hbm',
        # the kernel the last stamped Replay.sample launch ran (emb_replay_profile_report)
        'kernel': f'{gather_kernel} (Replay.sample)',
        'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        # frac counts read + written bytes (a copy moves two bytes per payload byte); the
        # HBM-READ fraction SURVEY 8d defines, (B*L*S / t) / 8e12, is read_frac
        'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
        'traffic_source': traffic_source,
        'read_frac': round(achieved / 2 / HBM_PEAK_GBS, 4),
        # context: the same bytes against the chip's measured copy rate (a gather cannot beat a copy)
        'frac_of_copy_rate': round(achieved / HBM_COPY_GBS, 4),
        'bytes_per_launch': algo_bytes, 'avg_launch_us': round(avg_s * 1e6, 2),
        'launches': launches, 'stamped_one_in': one_in, 'source': source,
        'headline_region': region,
    }

  # configs[2]: the write-back of the agent's latents over the sampled steps
  # (Replay.update, dreamerv3/agent.py:144-150), stamped like the gathers.
  writeback = None
  if wb_launches:
    wb_keys = [k for k in replay._keys if k.name in ('dyn/deter', 'dyn/stoch')]
    wb_bytes = 2 * B * args.prefetch * (T + args.context) * sum(k.rowbytes for k in wb_keys)
    wb_s = wb_ms / wb_launches / 1e3
    writeback = {
        'kernel': f'{wb_kernel} (Replay.update)', 'bytes_per_launch': wb_bytes,
        'avg_launch_us': round(wb_s * 1e6, 2), 'launches': wb_launches,
        'achieved': round(wb_bytes / wb_s / 1e9, 1), 'unit': 'GB/s',
        'frac': round(wb_bytes / wb_s / 1e9 / HBM_PEAK_GBS, 4),
        # r+w bytes over the launch time, like the gather's figure -- but the source of a
        # write-back is the batch the gather of the same train step just stored with plain
        # stores: most of its reads are served by L2 / Infinity Cache, only the writes must
        # reach HBM (write-only fraction = half of `frac`)
        'write_frac': round(wb_bytes / 2 / wb_s / 1e9 / HBM_PEAK_GBS, 4),
        'source': 'the agent\'s own (B, L, ...) output tensors, two sets used in turn (170 MB: partly cache-resident)',
    }

  # Outside the timed region, measured context (no credit): SURVEY 8d's
  # batches-per-launch sweep 1 / 4 / 8 / 16 / 64 (the reference's prefetch depth is 1, so
  # one batch of B=16 per launch is the faithful headline), and a plain
  # device-to-device copy of one batch's bytes out of the same pool — what a
  # kernel with no gather structure at all achieves on this box, read cold.
  if (roofline and rank == 0 and world == 1 and args.consec == 1 and not args.host_envs
      and not args.no_context):
    try:
      replay.profile(True, every=1)
      sweep = {}
      for per_launch in (1, 4, 8, 16, 64):       # B = 16, 64, 128, 256, 1024 sequences
        big = B * per_launch
        for _ in range(3):
          replay.sample(big, 'report')
        torch.cuda.synchronize(device)
        replay.profile_read(reset=True)
        # three rounds of 20 launches, the quietest one counts: these boxes are
        # shared, and one disturbed round once read 0.48 between 0.72 and 0.71
        best = None
        for _ in range(3):
          for _ in range(20):
            replay.sample(big, 'report')
          torch.cuda.synchronize(device)
          count, ms = replay.profile_read(reset=True)
          if count and (best is None or ms / count < best):
            best = ms / count
        if best is not None:
          us = best * 1e3
          gbs = 2 * big * L * S / (us * 1e-6) / 1e9
          sweep[str(per_launch)] = {
              'sequences': big, 'avg_launch_us': round(us, 2), 'achieved': round(gbs, 1),
              'frac': round(gbs / HBM_PEAK_GBS, 4), 'read_frac': round(gbs / 2 / HBM_PEAK_GBS, 4)}
      sweep['how'] = 'context: per size, the quietest of three rounds of 20 back-to-back launches'
      roofline['batches_per_launch_sweep'] = sweep
      roofline['plain_copy_same_bytes'] = plain_copy_reference(replay, B * L, device)
    except Exception as e:     # context only: never lose the headline over it
      roofline['batches_per_launch_sweep'] = {'error': str(e)[:200]}

  cpu = None
  if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == 'ppo' and not args.host_envs:
    cpu = cpu_baseline(args)

  # BASELINE configs[2] beside the headline (same command, own process): a short
  # DreamerV3-shaped run -- 1M-step uniform replay (78 GB of HBM), train_ratio
  # 32, 40 KB of latents per step written back -- so that its rates and the
  # roofline fractions of its 144 MB gathers and 84 MB write-backs are part of
  # every record.  Never part of `value`.
  workloads = None
  if (rank == 0 and world == 1 and not use_dist and args.workload == 'ppo' and not args.no_dreamer_leg
      and not args.host_envs and args.selector == 'uniform' and args.envs == 64
      and args.consec == 1 and args.prefetch == 1):
    del driver, env, stream, policy
    replay = None
    gc.collect()
    torch.cuda.empty_cache()
    workloads = {'dreamer': dreamer_leg(args)}

  env_actions = ('unmasked + reset (Env protocol, base.py:44-52)' if was_unmasked
                 else 'masked (driver.py:72-75)')
  if use_direct:
    # A wait inside the direct transport that gave up on a peer on ANY rank makes
    # the exchanged bytes of this run meaningless: the line says so.
    gave_up = torch.tensor([1.0 if direct_comm.timed_out() else 0.0], device=device)
    dist.all_reduce(gave_up, op=dist.ReduceOp.MAX)
    native['direct_timed_out_during_run'] = bool(gave_up.item())
    if rank == 0 and gave_up.item():
      print('bench.py: the direct transport ran into a wait time-out during this run', file=sys.stderr)
  if rank == 0:
    # Libraries that wrote to C stdio (RCCL's version banner) come out first, so
    # that the JSON line is the last line on stdout.
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps({
        'metric': 'env steps/sec + learner train-steps/sec, 64 envs 84x84x4 obs, 1/2/4/8 GPU',
        'value': round(env_steps / elapsed, 1),
        'unit': 'env_steps/s',
        'train_steps_per_s': round(train_steps / total_elapsed, 2),
        'n_gpus': world, 'rccl_ranks': rccl_ranks,
        'backend': (dist.get_backend() if use_dist else None),
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 5),
        # every timed region: exactly `steps` steps between two fences; `value` and
        # `ms_per_step` are the median region's, `train_steps_per_s` is over all of them
        'regions': {'n': n_regions,
                    'env_steps_per_s': {
                        'min': round(env_steps / max(region_times), 1),
                        'median': round(env_steps / elapsed, 1),
                        'max': round(env_steps / min(region_times), 1)},
                    'ms': [round(t * 1e3, 4) for t in region_times],
                    'train_steps': [c['train_steps'] for c in region_counts]},
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'u8', 'data': 'synthetic',
        'config': {
            'workload': (
                f'{"dreamerv3_1M_uniform" if args.workload == "dreamer" else "ppo_atari_pong"}_64env: '
                f'64 {"host (numpy, PCIe upload)" if args.host_envs else "device"} envs/GPU, 84x84x4 u8, '
                f'Replay(length={L}, capacity={args.capacity}, '
                f'{"uniform, latents written back" if args.workload == "dreamer" else "online"}'
                f'{", prioritized selector" if args.selector == "prioritized" else ""}), '
                f'B={B}, T={T}, train_ratio={args.train_ratio}, '
                f'{"lambda-return" if args.workload == "dreamer" else "GAE"}'),
            'envs_per_gpu': args.envs, 'global_envs': args.envs * world,
            **({'envs_per_worker': getattr(driver_ref[0], '_per_worker', None),
                'cpu_budget': round(emb.core.driver.cpu_budget(), 1)} if args.parallel_envs else {}),
            'batch': B, 'seq_len': L, 'batches_per_launch': args.prefetch, 'consec': args.consec,
            # 2: the train step's gather / scans / write-back on the learner's own HIP stream
            'streams': args.streams,
            **({'learner_cus': learner_cus, 'learner_first_cu': args.learner_first_cu} if args.streams == 2 else {}),
            # how the policy's actions reach the env: 'masked' = the reference's form (the default);
            # 'unmasked + reset' = the env declared that it ignores the action of an env it resets,
            # the Driver skips the masked copy and the Replay carries the masked pool write into its
            # next launch (carried publish): one dependent launch less per step
            'env_actions': {
                'value_measured_with': env_actions,
                **({'env_steps_per_s': {
                    env_actions: sustained['env_steps_per_s'],
                    ('masked (driver.py:72-75)' if was_unmasked else 'unmasked + reset (Env protocol, base.py:44-52)'):
                        other_env_actions['env_steps_per_s']},
                    'masked_over_unmasked': round(
                        (other_env_actions['env_steps_per_s'] / sustained['env_steps_per_s']) if was_unmasked
                        else (sustained['env_steps_per_s'] / other_env_actions['env_steps_per_s']), 3)}
                   if (other_env_actions is not None and sustained is not None) else {})},
            'kernargs': 'host' if os.environ.get('HIP_FORCE_DEV_KERNARG') == '0' else 'device',
            'cpus': PINNED,         # CPUs this process was pinned to (pin_cpus), None = scheduler's choice
            # the per-step Python of Driver / Replay / streams: Cython-compiled copies of the
            # same sources (embodied_amd/_compiled_finder.py) or the plain modules
            'host_modules': 'compiled' if 'embodied_amd.core.driver' in emb.compiled.loaded else 'python',
            'parallelism': (f'env-sharded x{world}, '
                            + ('per-rank Replay, ' if args.workload == 'dreamer'
                               else f'trajectory exchange {args.exchange} + ') +
                            f'{args.grad_numel * (2 if args.grad_dtype == "bf16" else 4) / 2**20:.1f} MiB '
                            f'{args.grad_dtype} grad all-reduce per train step'
                            + (' [bf16: LOWER precision than the reference\'s f32 pmean, opt.py:52-54]'
                               if args.grad_dtype == 'bf16' else '') + ' ('
                            + ('direct xGMI schedule, emb_direct_exchange' if use_direct else
                               f'RCCL, issued by {"emb_comm_exchange" if use_native else "torch.distributed"}')
                            + ')')
                           if use_dist else 'single',
        },
        'publishes': {'deferred': h_deferred, 'predicted': int(h_predicted), 'carried': int(h_carried),
                      'carried_inline': h_inline},
        'sustained': sustained,
        'roofline': roofline, 'cpu_baseline': cpu,
        **({'writeback': writeback} if writeback is not None else {}),
        **({'workloads': workloads} if workloads is not None else {}),
        **({'native_comm': native} if native is not None else {}),
        # N > 1, one place to read the scaling story from: who carried the timed path, every
        # transport's measured train-step exchange (host + end to end) on this job's own bytes,
        # and how often the ranks compared their exchange schedules at a fence (they agreed
        # every time, or the job would have ended)
        **({'transports': {
            'timed_path': (native or {}).get('timed_path', 'c10d'),
            'exchange': args.exchange, 'grad_dtype': args.grad_dtype,
            'grad_bytes': (grads.numel() * grads.element_size()) if grads is not None else 0,
            'exchange_step_us': {
                name: calls[key] for name, key, calls in (
                    ('c10d', f'c10d_{step_name}', (native or {}).get('per_call') or {}),
                    ('rccl', f'native_{step_name}', (native or {}).get('per_call') or {}),
                    ('direct', f'direct_{step_name}', ((native or {}).get('direct') or {}).get('per_call') or {}))
                if key in calls},
            'schedule_agreed_at_fences': schedule['fences']}} if use_dist else {}),
        **({'link_bound': native['per_train_step']['link_bound']}
           if native is not None and 'link_bound' in native.get('per_train_step', {}) else {}),
        **({'replicas_only': replicas_only} if replicas_only is not None else {}),
        **({'other_env_actions': other_env_actions} if other_env_actions is not None else {}),
        **({'fresh_batches': fresh_batches} if fresh_batches is not None else {}),
        **({'expected': expected} if expected is not None else {}),
    }), flush=True)
  if native_stuck:       # a collective of the check never returned: leave without the teardown
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)
  if use_dist:
    if comm is not None:
      comm.close()
    if native_comm is not None:
      native_comm.close()
    if direct_comm is not None:
      torch.cuda.synchronize(device)
      dist.barrier()           # nobody unmaps a buffer a peer's kernel may still be writing
      direct_comm.close()
    dist.destroy_process_group()


XGMI_LINK_GBS = 76.8         # one xGMI link, one direction (MI355X_MICROARCH.md: 153.6 GB/s bidirectional)


def scaling_expectation(world, grad_bytes, batch_bytes, sliced_share, train_period_us, gather_share=0.0,
                        gather_bytes=None):
  """DESIGN.md 5's link budget as a formula, so that a measured curve can be held
  against it: per train step and rank, one way, a direct all-reduce moves
  2*(n-1)/n * G bytes; on the share of train steps that exchange DP slices the
  all-to-all moves (n-1)/n * B*L*S more, on the share that all-gathers (n-1) *
  `gather_bytes` (default: the whole batch B*L*S) -- all over the (n-1)
  point-to-point links a rank has to its peers.  `train_period_us` is the
  period at which one rank issues train steps with the collectives off (the
  `replicas_only` loop of the same run).  x = speed-up over one such rank."""
  n = world
  gather_bytes = batch_bytes if gather_bytes is None else gather_bytes
  one_way = (2 * (n - 1) / n * grad_bytes + sliced_share * (n - 1) / n * batch_bytes
             + gather_share * (n - 1) * gather_bytes)
  out = {'bytes_one_way_per_train_step': round(one_way), 'links': n - 1,
         'train_period_us_collectives_off': round(train_period_us, 1), 'replicas_only_x': float(n)}
  for pct in (100, 60, 40):
    link_us = one_way / ((n - 1) * XGMI_LINK_GBS * 1e9 * pct / 100) * 1e6 if n > 1 else 0.0
    out[f'link_time_us_at_{pct}pct'] = round(link_us, 1)
    out[f'link_bound_x_at_{pct}pct'] = round(n * min(1.0, train_period_us / max(link_us, 1e-9)), 2)
  return out


def _fault(kind, rank):
  """Test hook (tests/test_gpu_bench_launcher.py): `EMB_BENCH_FAULT=kind:rank` makes
  ONE rank misbehave inside the self-check of the transports, the way a node
  might on first contact -- `direct_open:R` (rank R cannot make / map its hipIpc
  buffer), `direct_stuck:R` (rank R never arrives at one direct collective),
  `direct_wrong:R` (rank R's direct all-to-all delivers wrong bytes), `native_open:R`
  (rank R's RCCL communicator set-up fails before it starts).  The job must end
  with rc 0 and a valid line on the transport that is left."""
  spec = os.environ.get('EMB_BENCH_FAULT', '')
  return spec == f'{kind}:{rank}'


def native_comm_check(rank, world, device, grad_numel, grad_dtype, slice_bytes, gather_bytes, seconds=120.0):
  """Ranks only, before the timed regions (no part of `value`): the library's own
  transports on the GPUs of this job -- the RCCL entry points (`emb_comm_*`) and
  the direct xGMI schedule (`emb_direct_*`, include/embodied_hip.h) -- checked
  against torch.distributed on the same bytes (all-gather, DP-slice all-to-all,
  f32 and bf16 gradient all-reduce) and timed, per call and per train-step
  exchange, beside the process group's own route.

  Built so that a transport that FAILS on some rank costs this check seconds and
  the job nothing: (1) every torch.distributed collective of the check runs on a
  group of its own, unconditionally and in one fixed order on every rank -- the
  references are computed first, a transport's own operations (which may raise,
  time out or deliver garbage on one rank only) contain no process-group call;
  (2) after each phase the ranks agree (MIN all-reduce) whether the transport is
  still healthy EVERYWHERE, and all take the same branch; (3) the whole check
  runs under a watchdog: if a collective never returns the job goes on with
  torch.distributed.  Returns (report, stuck, NativeComm or None, DirectComm or
  None): a communicator only if every rank passed."""
  import threading
  import torch.distributed as dist
  from embodied_amd import distributed as D
  result = {'status': 'timeout'}
  kept = {}
  backend = dist.get_backend()
  group = dist.new_group(backend=backend)
  pg = group
  staged = backend != 'nccl'      # gloo moves host memory: the reference side goes through the CPU
  reps = 10 if staged else 100    # (a loopback test transport: the figures mean nothing there)

  def share(data):
    box = [data]
    dist.broadcast_object_list(box, src=0, group=group, **({} if staged else {'device': device}))
    return box[0]

  def share_all(data):
    box = [None] * world
    dist.all_gather_object(box, data, group=group)
    return box

  def agree(ok):
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() == 1.0)

  def ref_all_gather(out, part):
    if not staged:
      return dist.all_gather_into_tensor(out, part, group=group)
    parts = [torch.empty(part.shape, dtype=part.dtype) for _ in range(world)]
    dist.all_gather(parts, part.cpu(), group=group)
    out.copy_(torch.cat(parts))

  def ref_all_to_all(out, whole):
    if not staged:
      return pg.alltoall_base(out, whole, [], [])
    host = torch.empty(whole.shape, dtype=whole.dtype)
    pg.alltoall_base(host, whole.cpu(), [], []).wait()
    out.copy_(host)
    return D._Finished()

  def ref_all_reduce(buf, mean=False):
    if not staged:
      return dist.all_reduce(buf, op=dist.ReduceOp.AVG if mean else dist.ReduceOp.SUM, group=group)
    host = buf.float().cpu()
    dist.all_reduce(host, group=group)
    buf.copy_((host / world if mean else host).to(buf.dtype))

  def timed(routes, failed):
    """host_us / total_us per call of every route.  The barrier in front of each
    route is taken by every rank whatever happened to its own calls."""
    costs = {}
    for name, call in routes.items():
      ok = not failed[0]
      try:
        for _ in range(reps // 10 if ok else 0):
          call()
        torch.cuda.synchronize(device)
      except Exception as e:
        failed[0] = f'{name}: {type(e).__name__}: {e}'[:300]
        ok = False
      dist.barrier(group=group)
      if not ok:
        continue
      try:
        t0 = time.perf_counter()
        for _ in range(reps):
          call()
        t1 = time.perf_counter()
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        costs[name] = {'host_us': round((t1 - t0) * 1e6 / reps, 2), 'total_us': round((t2 - t0) * 1e6 / reps, 2)}
      except Exception as e:
        failed[0] = f'{name}: {type(e).__name__}: {e}'[:300]
    return costs

  def body():
    gen = torch.Generator(device=device)
    gen.manual_seed(977 + rank)
    block = max(16, slice_bytes // 16 * 16)
    whole_block = max(16, gather_bytes // 16 * 16)          # one rank's packed batch (the all-gather's unit)
    flat = torch.randint(0, 256, (world * block,), dtype=torch.uint8, device=device, generator=gen)
    mine = torch.randint(0, 256, (whole_block,), dtype=torch.uint8, device=device, generator=gen)
    ints = {name: torch.randint(-8, 9, (1 << 20,), device=device, generator=gen).to(dtype)
            for name, dtype in (('f32', torch.float32), ('bf16', torch.bfloat16))}
    noise = {name: torch.randn(1 << 20, device=device, generator=gen).to(dtype)
             for name, dtype in (('f32', torch.float32), ('bf16', torch.bfloat16))}
    # ---- the references, on the process group: every rank, one order, nothing else in between
    ref = {'all_gather': torch.empty(world * whole_block, dtype=torch.uint8, device=device),
           'all_to_all': torch.empty_like(flat)}
    ref_all_gather(ref['all_gather'], mine)
    ref_all_to_all(ref['all_to_all'], flat).wait()
    for name in ints:
      ref[f'sum_{name}'] = ints[name].clone()
      ref_all_reduce(ref[f'sum_{name}'])
      ref[f'mean_{name}'] = noise[name].clone()
      ref_all_reduce(ref[f'mean_{name}'], mean=True)
    torch.cuda.synchronize(device)
    grads = torch.zeros(max(grad_numel, 1), dtype=grad_dtype, device=device)
    recv = torch.empty_like(flat)
    everyone = torch.empty(world * whole_block, dtype=torch.uint8, device=device)
    report = {'ranks': world, 'all_reduce_bytes': grads.numel() * grads.element_size(),
              'all_to_all_bytes': flat.numel(), 'all_gather_bytes_per_rank': whole_block}

    def close(a, b, dtype):
      tol = 1e-5 if dtype == 'f32' else 2e-2
      return bool(torch.allclose(a.float(), b.float(), rtol=tol, atol=tol))

    # ---- the process group's own route, timed (the figure the other two stand beside)
    failed = [None]
    report['c10d'] = {'per_call': timed({
        'c10d_all_reduce': lambda: pg.allreduce([grads]),
        'c10d_all_to_all': lambda: ref_all_to_all(recv, flat),
        'c10d_all_gather': lambda: ref_all_gather(everyone, mine),
        'c10d_exchange_step': lambda: (ref_all_to_all(recv, flat).wait(), pg.allreduce([grads]).wait()),
        'c10d_exchange_gather_step': lambda: (ref_all_gather(everyone, mine), pg.allreduce([grads]).wait()),
    }, failed)}

    # ---- RCCL through the library's own entry points (emb_comm_*)
    comm, error, checks, ident = None, None, {}, None
    try:
      if _fault('native_open', rank):
        raise RuntimeError('injected: this rank cannot set up its RCCL communicator')
      ident = D.NativeComm.unique_id()        # every rank: binds the RCCL symbols (rank 0's id is the one used)
    except Exception as e:
      error = f'{type(e).__name__}: {e}'[:300]
    if agree(error is None):      # (ncclCommInitRank blocks until every rank has joined: all or nobody)
      ident = share(ident)
      try:
        comm = D.NativeComm(rank, world, device, ident=ident)
        kept['native'] = comm
        checks['all_gather'] = bool(torch.equal(comm.all_gather(mine), ref['all_gather']))
        checks['all_to_all'] = bool(torch.equal(comm.all_to_all(flat), ref['all_to_all']))
        for name in ints:
          a = ints[name].clone()
          comm.all_reduce(a, mean=False)
          checks[f'all_reduce_sum_{name}'] = bool(torch.equal(a, ref[f'sum_{name}']))
          a = noise[name].clone()
          comm.all_reduce(a, mean=True)
          checks[f'all_reduce_mean_{name}'] = close(a, ref[f'mean_{name}'], name)
        a, b = torch.zeros_like(everyone), noise['f32'].clone()
        comm.exchange(mine, a, b, gather=True)
        comm.wait()
        checks['exchange_gather'] = bool(torch.equal(a, ref['all_gather'])) and close(b, ref['mean_f32'], 'f32')
      except Exception as e:
        error = f'{type(e).__name__}: {e}'[:300]
    passed = agree(error is None and bool(checks) and all(checks.values()))
    failed = [None if passed else 'the check did not pass on every rank']
    costs = timed({
        'native_all_reduce': lambda: comm.all_reduce(grads, mean=True),
        'native_all_to_all': lambda: comm.all_to_all(flat, recv),
        'native_all_gather': lambda: comm.all_gather(mine, everyone),
        # one train step's worth, own stream, with the wait of the previous one
        'native_exchange_step': lambda: (comm.wait(), comm.exchange(flat, recv, grads)),
        'native_exchange_gather_step': lambda: (comm.wait(), comm.exchange(mine, everyone, grads, gather=True)),
    }, failed)
    passed = agree(passed and failed[0] is None)
    report.update({
        'status': ('ok' if passed else 'error' if error else 'mismatch' if checks and not all(checks.values())
                   else 'failed on another rank'),
        'transport': os.environ.get('EMB_RCCL_LIB') or 'rccl', 'checks': checks,
        **({'error': error or failed[0]} if (error or failed[0]) else {}),
        'per_call': {**report['c10d']['per_call'], **costs}})

    # ---- the direct xGMI schedule (emb_direct_*): every rank writes all n-1 peers at once through
    # hipIpc pointers.  Its failure never takes the RCCL path down.
    direct, derror, dchecks = None, None, {}
    try:
      direct = D.DirectComm(
          rank, world, device,
          max_grad_bytes=(1 << 50) if _fault('direct_open', rank) else max(grads.numel() * grads.element_size(), 4 << 20),
          max_slice_bytes=max(2 * block, whole_block + 4096, 1 << 20),      # (a packed batch pads its keys)
          share_all=share_all, timeout_ms=3000)     # a peer that does not answer costs this check seconds, not the job
      kept['direct'] = direct
    except Exception as e:
      derror = f'{type(e).__name__}: {e}'[:300]
    if agree(direct is not None):
      try:
        if _fault('direct_stuck', rank):
          raise RuntimeError('injected: this rank never arrives at the direct collectives')
        got = direct.all_to_all(flat)
        if _fault('direct_wrong', rank):
          got = got ^ 1
        dchecks['all_to_all'] = bool(torch.equal(got, ref['all_to_all']))
        dchecks['all_gather'] = bool(torch.equal(direct.all_gather(mine), ref['all_gather']))
        for name in ints:
          a = ints[name].clone()
          direct.all_reduce(a, mean=False)
          dchecks[f'all_reduce_sum_{name}'] = bool(torch.equal(a, ref[f'sum_{name}']))
          a = noise[name].clone()
          direct.all_reduce(a, mean=True)
          dchecks[f'all_reduce_mean_{name}'] = close(a, ref[f'mean_{name}'], name)
        a, b = torch.zeros_like(everyone), noise['f32'].clone()
        direct.exchange(mine, a, b, gather=True)
        direct.wait()
        dchecks['exchange_gather'] = bool(torch.equal(a, ref['all_gather'])) and close(b, ref['mean_f32'], 'f32')
        direct.check()            # synchronises; raises if a wait inside a kernel gave up on a peer
      except Exception as e:
        derror = f'{type(e).__name__}: {e}'[:300]
      dpassed = agree(derror is None and len(dchecks) == 7 and all(dchecks.values()))
      failed = [None if dpassed else 'the check did not pass on every rank']
      dcosts = timed({
          'direct_all_reduce': lambda: direct.all_reduce(grads, mean=True),
          'direct_all_to_all': lambda: direct.all_to_all(flat, recv),
          'direct_all_gather': lambda: direct.all_gather(mine, everyone),
          'direct_exchange_step': lambda: (direct.wait(), direct.exchange(flat, recv, grads)),
          'direct_exchange_gather_step': lambda: (direct.wait(), direct.exchange(mine, everyone, grads, gather=True)),
      }, failed)
      timed_out = False
      try:
        if dpassed:
          direct.wait()
        timed_out = direct.timed_out()
      except Exception as e:
        failed[0] = failed[0] or f'{type(e).__name__}: {e}'[:300]
        timed_out = True
      dpassed = agree(dpassed and failed[0] is None and not timed_out)
      report['direct'] = {
          'status': ('ok' if dpassed else 'error' if derror else 'mismatch' if dchecks and not all(dchecks.values())
                     else 'failed on another rank'),
          'checks': dchecks, 'timed_out': timed_out,
          **({'error': derror or failed[0]} if (derror or failed[0]) else {}),
          **({'per_call': dcosts} if dcosts else {})}
    else:
      report['direct'] = {'status': 'error' if derror else 'failed on another rank',
                          **({'error': derror} if derror else {})}
    # nobody closes (unmaps) a buffer that a peer's kernel may still be writing
    torch.cuda.synchronize(device)
    dist.barrier(group=group)
    return report

  def guarded():
    torch.cuda.set_device(device)
    try:
      local = body()
    except Exception as e:       # (a bug of the check itself; the transports' own failures are handled above)
      local = {'status': 'error', 'error': f'{type(e).__name__}: {e}'[:300]}
    result.clear()
    result.update(local)

  worker = threading.Thread(target=guarded, name='native_comm_check', daemon=True)
  worker.start()
  worker.join(seconds)
  stuck = worker.is_alive()
  report = dict(result)
  usable = kept.get('native') if (not stuck and report.get('status') == 'ok') else None
  if kept.get('native') is not None and usable is None and not stuck:
    kept['native'].close()
  direct = kept.get('direct') if (not stuck and (report.get('direct') or {}).get('status') == 'ok') else None
  if kept.get('direct') is not None and direct is None and not stuck:
    try:
      torch.cuda.synchronize(device)
      kept['direct'].close()
    except Exception:
      pass
  if direct is not None:
    direct.set_timeout(600_000)       # the job itself: a collective watchdog, not the check's three seconds
  return report, stuck, usable, direct


def dreamer_leg(args):
  """`python bench.py --workload dreamer` for a few seconds in a process of its
  own (HIP state, allocator and CPU placement as in a stand-alone run), twice:
  with the replay-context latents sampled as the shipped agent consumes them
  (`--context-only`: (B, K, ...), dreamerv3/agent.py:322-331) and with all L steps
  of them gathered (`full_gather`, what the reference's Replay.sample returns and
  rounds 1-4 measured).  Returns the part of the lines that a reader of the PPO
  record needs."""
  import subprocess

  def run(*extra):
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'dreamer', '--no-dreamer-leg',
           '--steps', str(args.dreamer_leg_steps), '--warmup', '200', '--sustained-seconds', '3',
           '--no-cpu-baseline', '--no-context', *extra]
    began = time.perf_counter()
    try:
      res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
      lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
      if res.returncode or not lines:
        return {'error': (res.stderr or res.stdout)[-300:], 'returncode': res.returncode}
      line = json.loads(lines[-1])
    except Exception as e:
      return {'error': f'{type(e).__name__}: {e}'[:300]}
    roof, wb = line.get('roofline') or {}, line.get('writeback') or {}
    return {
        'workload': line['config']['workload'],
        'command': ' '.join(['python', 'bench.py'] + cmd[2:]),
        'env_steps_per_s': line['value'], 'train_steps_per_s': line['train_steps_per_s'],
        'ms_per_step': line['ms_per_step'], 'steps': line['steps'],
        'sustained': line.get('sustained'),
        'gather': {k: roof.get(k) for k in (
            'kernel', 'avg_launch_us', 'bytes_per_launch', 'achieved', 'frac', 'read_frac', 'launches')},
        'writeback': {k: wb.get(k) for k in (
            'kernel', 'avg_launch_us', 'bytes_per_launch', 'achieved', 'frac', 'launches')},
        'wall_s': round(time.perf_counter() - began, 1),
    }

  leg = run('--context-only')
  if 'error' in leg:
    return leg
  leg['latents_sampled'] = f'context only: (B, K={args.context}, ...) of dyn/ (dreamerv3/agent.py:322-331)'
  full = run()
  full['latents_sampled'] = 'all L steps (the reference\'s Replay.sample)'
  leg['full_gather'] = full
  return leg


def plain_copy_reference(replay, rows, device, iters=200):
  """Back-to-back torch copies of `rows` image rows from rotating (cold) places
  of the replay's image pool into one output buffer: us per copy (events around
  the loop, i.e. throughput — a tight loop keeps the GPU busy) and r+w GB/s."""
  key = max(replay._keys, key=lambda k: k.rowbytes)
  nbytes = rows * key.rowbytes
  pool = key.pool
  slots = max(1, pool.numel() // nbytes - 1)
  out = torch.empty(nbytes, dtype=torch.uint8, device=device)
  srcs = [pool[(i * 7919 % slots) * nbytes:][:nbytes] for i in range(iters)]
  for src in srcs[:10]:
    out.copy_(src)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for src in srcs:
    out.copy_(src)
  e1.record()
  torch.cuda.synchronize(device)
  us = e0.elapsed_time(e1) / iters * 1e3
  return {'bytes_one_way': nbytes, 'us_per_copy': round(us, 2),
          'achieved': round(2 * nbytes / us / 1e3, 1),
          'frac': round(2 * nbytes / us / 1e3 / HBM_PEAK_GBS, 4),
          'how': 'torch copy_ of the image key\'s share of one batch, cold source, '
                 'throughput over back-to-back launches'}


def pmc_traffic(algo_bytes):
  """HBM bytes per gather launch from the committed PMC passes (counters cannot
  be read from inside the process): the newest profiles/r*_pmc_gather.json whose
  workload matches this run, else None."""
  import glob
  here = os.path.dirname(os.path.abspath(__file__))
  for path in sorted(glob.glob(os.path.join(here, 'profiles', 'r*_pmc_gather.json')), reverse=True):
    try:
      with open(path) as f:
        rec = json.load(f)
      if rec['algorithmic_bytes_per_launch'] == algo_bytes:
        return rec['traffic_bytes_per_launch'], os.path.relpath(path, here)
    except Exception:
      continue
  return None, None


def cpu_baseline(args):
  """The numpy oracle of the same step on this host, single thread: serial
  Driver over 64 host synthetic envs + oracle Replay.add/sample + numpy GAE.
  Same replay capacity as the GPU leg (filled first, untimed), then
  ~args.cpu_seconds of timed work."""
  from oracle import np_oracle
  from embodied_amd.envs import synthetic
  L, B, T = args.length + args.context, args.batch, args.length
  envs = [synthetic.HostSyntheticEnv(e) for e in range(args.envs)]
  drv = np_oracle.Driver(envs)
  rep = np_oracle.Replay(L, args.capacity, 1024, online=True, seed=0)
  drv.on_step(rep.add)
  gen = np.random.default_rng(0)
  val = gen.standard_normal((B, L)).astype(np.float32)

  def policy(carry, obs):
    n = len(obs['is_first'])
    _ = np.ascontiguousarray(obs['image'].transpose(0, 3, 1, 2)).astype(np.float32) * (1 / 255)
    return carry, {'action': gen.integers(0, 6, n).astype(np.int32)}, {}

  should_train = Ratio(args.train_ratio / (B * T))
  env_steps = train_steps = 0
  fill_begin = time.perf_counter()
  while len(rep) < args.capacity and time.perf_counter() - fill_begin < 60:
    drv.step(policy)
  filled = len(rep)
  begin = time.perf_counter()
  while time.perf_counter() - begin < args.cpu_seconds:
    drv.step(policy)
    env_steps += args.envs
    for _ in range(should_train(env_steps)):
      batch = rep.sample(B)
      np_oracle.gae(batch['reward'], val, batch['is_last'], batch['is_terminal'])
      train_steps += 1
  took = time.perf_counter() - begin
  return {
      'value': round(env_steps / took, 1), 'unit': 'env_steps/s', 'cores': 1,
      'kind': 'port',
      'train_steps_per_s': round(train_steps / took, 3),
      'sample': f'{env_steps} env steps / {train_steps} train steps in {took:.1f}s, '
                f'numpy oracle, replay filled to {filled} of {args.capacity} items, '
                'same envs/shapes/ratio',
      'host_cpus': os.cpu_count(),
  }


if __name__ == '__main__':
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  main()
