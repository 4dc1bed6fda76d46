"""embodied_amd — MI355X-native Driver / Replay / return-scan hot path behind
the reference's Python interfaces."""
__version__ = '0.1.0'

# (Importing this package changes nothing about the process: no environment
# variables, no CPU affinity.  A host program that wants the cheaper launches of
# host-resident kernel arguments sets HIP_FORCE_DEV_KERNARG=0 itself before HIP
# starts, as bench.py does -- INTEGRATION.md; the library serves both placements.)

from . import _compiled_finder
compiled = _compiled_finder.install()   # compiled copies of the hot host modules, if built and fresh

from . import _lib  # raises if libembodied_hip.so is missing: no CPU fallback
from ._lib import configure

from .space import Space
from .core.base import Agent, Env, Stream
from .core.driver import Driver
from .core.agents import RandomAgent
from .core.replay import Replay
from .core import limiters
from .core import selectors
from .core import streams
from .core import wrappers
from .core.wrappers import Wrapper
from .core import replay
from . import scans
from . import ops
from . import envs
from . import utils
from . import run
from . import distributed
from .utils import Counter, LocalClock
from .distributed import GlobalClock
# `embodied.clock.*` by name (embodied/core/__init__.py:4-5,11): the two clocks and
# `setup`; the replicas meet on a torch.distributed group instead of an RPC server.
import types as _types
clock = _types.SimpleNamespace(
    LocalClock=LocalClock, GlobalClock=GlobalClock, setup=distributed.clock_setup)
