"""`embodied.clock` by name (embodied/core/clock.py): `LocalClock`, `GlobalClock`
and `setup`.

The reference's GlobalClock asks an RPC server on replica 0 (clock.py:11-74,
`setup(is_server, replica, replicas, port, addr)` starts it); this package's
replicas are the ranks of a `torch.distributed` job, and `GlobalClock` takes its
lockstep decision with one MAX all-reduce over that group
(`embodied_amd.distributed.GlobalClock`, same rules as the server's `should`).
`setup` therefore has nothing to start: it checks that the replica layout it is
told about is the process group's and returns.
"""
from ..utils import LocalClock


def GlobalClock(every, first=False, group=None, device=None):
  """clock.py:77-94.  A LocalClock without a process group of more than one rank."""
  from ..distributed import GlobalClock as _GlobalClock
  return _GlobalClock(every, first, group=group, device=device)


def setup(is_server=False, replica=0, replicas=1, port=None, addr=None):
  """clock.py:11-24.  With one replica: nothing, as in the reference.  With more,
  the decisions travel over the `torch.distributed` group the job has
  initialised (`embodied_amd.distributed.init`): `port` / `addr` are not used,
  `replica` / `replicas` must be that group's rank and size."""
  if replicas <= 1:
    return
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()):
    raise RuntimeError(
        'clock.setup: more than one replica needs an initialised torch.distributed '
        'process group (embodied_amd.distributed.init)')
  if dist.get_world_size() != replicas or dist.get_rank() != replica:
    raise ValueError(
        f'clock.setup(replica={replica}, replicas={replicas}) but the process group says '
        f'rank {dist.get_rank()} of {dist.get_world_size()}')
