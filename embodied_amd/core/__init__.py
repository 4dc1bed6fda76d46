from .base import Agent, Env, Stream
from . import limiters
from . import selectors
