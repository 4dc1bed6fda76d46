from .base import Agent, Env, Stream
from .driver import Driver
from .agents import RandomAgent
from .replay import Replay
from . import limiters
from . import selectors
from . import streams
