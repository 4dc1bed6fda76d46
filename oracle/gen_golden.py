"""Generate tests/golden/*.npz by running the REAL reference modules.

TEST INFRASTRUCTURE, build container only (needs /root/reference).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Each scenario in tests/scenarios.py is executed against
`/root/reference/embodied/core` (imported unmodified through oracle/refload.py
and the `elements`/`portal` stand-ins in oracle/shims/).  Only data — scenario
outputs — is written; no reference source is copied.  Values that depend on the
stand-ins rather than on reference code: the first 16 bytes of each `stepid`
(chunk UUID; the stand-in's debug counter), nothing else.
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True

from tests import adapters, scenarios  # noqa: E402


def main():
  ns = adapters.reference_ns()
  outdir = ROOT / 'tests' / 'golden'
  outdir.mkdir(exist_ok=True)
  for name, fn in {**scenarios.SCENARIOS, **scenarios.HOST_SCENARIOS}.items():
    result = fn(ns)
    result = {k.replace('/', '__'): np.asarray(v) for k, v in result.items()}
    np.savez_compressed(outdir / f'{name}.npz', **result)
    size = (outdir / f'{name}.npz').stat().st_size
    print(f'{name:28s} {len(result):3d} arrays {size:7d} bytes')


if __name__ == '__main__':
  main()
