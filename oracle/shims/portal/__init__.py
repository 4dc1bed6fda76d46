"""Stand-in for the *used surface* of the un-vendored `portal` package.

TEST INFRASTRUCTURE ONLY (see oracle/shims/elements/__init__.py).  Only what
`embodied/core` needs at import time plus `Thread`; RPC classes are absent.
"""
import threading


class _Context:
  running = True


class Thread:

  def __init__(self, fn, *args, start=False, name=None):
    self._thread = threading.Thread(
        target=fn, args=(_Context(), *args) if False else args, daemon=True)
    if start:
      self.start()

  def start(self):
    self._thread.start()

  def join(self, timeout=None):
    self._thread.join(timeout)

  def kill(self):
    pass


class Process:

  def __init__(self, fn, *args, start=False):
    raise NotImplementedError(
        'portal.Process stand-in: golden vectors use Driver(parallel=False)')


class Client:
  def __init__(self, *a, **k):
    raise NotImplementedError


class Server:
  def __init__(self, *a, **k):
    raise NotImplementedError
