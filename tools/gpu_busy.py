"""GPU busy fraction of a rocprofv3 --kernel-trace CSV over the middle half of
its timeline (sum of kernel durations / wall), and the mean gap between
consecutive kernels:   python tools/gpu_busy.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1], newline='') as f:
  for r in csv.DictReader(f):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
lo, hi = len(rows) // 4, 3 * len(rows) // 4
mid = rows[lo:hi]
wall = mid[-1][1] - mid[0][0]
busy = sum(e - s for s, e, _ in mid)
gaps = [max(0, mid[i + 1][0] - mid[i][1]) for i in range(len(mid) - 1)]
envs = sum(1 for _, _, n in mid if 'synth_env' in n)
print(f'kernels {len(mid)}  wall {wall / 1e6:.2f} ms  busy {busy / wall:.3f}  '
      f'mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us  per env step: wall {wall / max(envs, 1) / 1e3:.2f} us, '
      f'kernels {busy / max(envs, 1) / 1e3:.2f} us')
