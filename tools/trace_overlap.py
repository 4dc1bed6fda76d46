"""Timeline view of a rocprofv3 --kernel-trace CSV: per queue, the busy time and
how much of it overlaps kernels of another queue; the sequence of one window.
    python tools/trace_overlap.py <kernel_trace.csv> [skip_fraction]
"""
import collections
import csv
import re
import sys

path = sys.argv[1]
rows = []
with open(path, newline='') as f:
  for row in csv.DictReader(f):
    name = re.sub(r'emb::\(anonymous namespace\)::', '', row['Kernel_Name'])
    name = re.sub(r'\(.*', '', name).replace('void ', '')[:48]
    rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), row.get('Queue_Id', '?'), name))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
rows = rows[int(len(rows) * skip):]            # the steady state at the end of the run
t0, t1 = rows[0][0], max(r[1] for r in rows)
queues = sorted({r[2] for r in rows})
print(f'{len(rows)} dispatches over {(t1 - t0) / 1e3:.0f} us on queues {queues}')
busy = collections.Counter()
per_kernel = collections.defaultdict(lambda: [0, 0])
for s, e, q, n in rows:
  busy[q] += e - s
  per_kernel[(q, n)][0] += 1
  per_kernel[(q, n)][1] += e - s
# union of all kernels = time with anything running
events = sorted([(s, 1) for s, e, q, n in rows] + [(e, -1) for s, e, q, n in rows])
depth, last, union, multi = 0, t0, 0, 0
for t, d in events:
  if depth >= 1:
    union += t - last
  if depth >= 2:
    multi += t - last
  depth += d
  last = t
print(f'any kernel running {union / (t1 - t0):.3f} of the time, two or more {multi / (t1 - t0):.3f}; '
      f'sum of kernel durations {sum(busy.values()) / (t1 - t0):.3f}')
for q in queues:
  print(f'queue {q}: busy {busy[q] / (t1 - t0):.3f}')
for (q, n), (c, ns) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
  print(f'  q{q} {n:50s} {c:7d} x {ns / c / 1e3:7.2f} us = {ns / (t1 - t0):.3f}')
mid = len(rows) // 2
print('a window of the timeline (start us, duration us, queue, kernel):')
base = rows[mid][0]
for s, e, q, n in rows[mid:mid + 40]:
  print(f'  {(s - base) / 1e3:8.2f} {(e - s) / 1e3:7.2f}  q{q}  {n}')
