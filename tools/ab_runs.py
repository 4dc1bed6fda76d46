"""Alternating runs of bench.py variants on ONE box (boxes differ by +-10 %, and a
box's host by as much from minute to minute): every round runs each variant once,
the table gives every run and the variants' medians and ratios.

    python tools/ab_runs.py OUT.txt ROUNDS "name::extra bench flags" "name::flags" ...
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = ['--no-cpu-baseline', '--no-context', '--no-dreamer-leg', '--sustained-seconds', '5']


def main():
  out, rounds, specs = sys.argv[1], int(sys.argv[2]), [s.split('::', 1) for s in sys.argv[3:]]
  rows = {name: [] for name, _ in specs}
  lines = [f'# python bench.py {" ".join(BASE)} + the variant\'s flags; {rounds} alternating rounds on one box',
           '# columns: variant | round | value (timed region) | sustained env steps/s | us per step | gather us | closing_fence us '
           '(milliseconds: the GPU sets the pace; tens of microseconds: the host does)']
  for r in range(1, rounds + 1):
    for name, flags in specs:
      res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *BASE, *flags.split()],
                           capture_output=True, text=True, cwd='/tmp')
      try:
        d = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
        s = d['sustained']
        rows[name].append(s['env_steps_per_s'])
        lines.append(f'{name} | {r} | {d["value"]:.0f} | {s["env_steps_per_s"]:.0f} | {s["ms_per_step"] * 1e3:.2f} | '
                     f'{s["gather_avg_us"]} | {s["closing_fence_us"]}')
      except Exception as e:
        lines.append(f'{name} | {r} | FAILED {e}')
      print(lines[-1], flush=True)
  first = specs[0][0]
  for name, _ in specs:
    if rows[name]:
      med = statistics.median(rows[name])
      lines.append(f'# {name}: median sustained {med:.0f} env steps/s over {len(rows[name])} runs'
                   + (f' = {med / statistics.median(rows[first]):.3f} x "{first}"' if name != first and rows[first] else ''))
      print(lines[-1], flush=True)
  with open(out, 'w') as f:
    f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
  main()
