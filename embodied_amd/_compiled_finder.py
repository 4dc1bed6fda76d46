"""Import hook for the Cython-compiled copies of the package's hot host modules.

`build.py` compiles the modules listed in its COMPILED table (the per-step
Python of Driver / Replay / streams / scans / ops ...) from their `.py` sources,
unchanged, into `embodied_amd/_compiled/<dotted name>.so` and writes
`manifest.json` with the sha-256 of every source it compiled.  This finder puts
a compiled module in front of its source ONLY while that digest still matches
the source file next to it -- a stale binary can never shadow edited code -- and
never at all with EMB_PURE_PYTHON=1.  Missing directory, missing compiler,
mismatching digest: the plain `.py` is imported, same behaviour, ~10-15 % slower
vectorised step."""
import hashlib
import importlib.abc
import importlib.machinery
import importlib.util
import json
import os
import pathlib
import sys

HERE = pathlib.Path(__file__).resolve().parent
DIR = HERE / '_compiled'


def _manifest(directory):
  try:
    return json.loads((directory / 'manifest.json').read_text())
  except (OSError, ValueError):
    return {}


class Finder(importlib.abc.MetaPathFinder):

  def __init__(self, directory=DIR, root=HERE.parent):
    self.table = {}
    self.loaded = []
    if os.environ.get('EMB_PURE_PYTHON') == '1':
      return
    for name, entry in _manifest(directory).items():
      binary, source = directory / entry['file'], root / entry['source']
      try:
        fresh = hashlib.sha256(source.read_bytes()).hexdigest() == entry['sha256']
      except OSError:
        fresh = False
      if fresh and binary.exists():
        self.table[name] = (binary, source)

  def find_spec(self, fullname, path=None, target=None):
    found = self.table.get(fullname)
    if found is None:
      return None
    binary, source = found
    loader = importlib.machinery.ExtensionFileLoader(fullname, str(binary))
    spec = importlib.util.spec_from_file_location(fullname, str(binary), loader=loader)
    self.loaded.append(fullname)
    return spec


def install():
  for finder in sys.meta_path:
    if isinstance(finder, Finder):
      return finder
  finder = Finder()
  sys.meta_path.insert(0, finder)
  return finder
