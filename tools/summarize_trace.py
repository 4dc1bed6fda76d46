"""Median duration per (kernel, grid) from a rocprofv3 --kernel-trace CSV:
    python tools/summarize_trace.py <kernel_trace.csv> [min_dispatches]
"""
import collections
import csv
import re
import statistics
import sys

path = sys.argv[1]
least = int(sys.argv[2]) if len(sys.argv) > 2 else 20
groups = collections.defaultdict(list)
with open(path, newline='') as f:
  for row in csv.DictReader(f):
    name = re.sub(r'emb::\(anonymous namespace\)::', '', row['Kernel_Name'])
    name = re.sub(r'\(.*', '', name).replace('void ', '')
    grid = int(row['Grid_Size_X']) * int(row['Grid_Size_Y']) * int(row['Grid_Size_Z'])
    groups[(name, grid, int(row['Workgroup_Size_X']))].append(
        int(row['End_Timestamp']) - int(row['Start_Timestamp']))
print('kernel,grid_threads,workgroup,dispatches,median_ns,p10_ns,p90_ns')
for (name, grid, wg), durations in sorted(groups.items()):
  if len(durations) < least or name.startswith('at::') or 'rocclr' in name:
    continue
  d = sorted(durations)
  print(f'"{name}",{grid},{wg},{len(d)},{int(statistics.median(d))},'
        f'{d[len(d) // 10]},{d[len(d) * 9 // 10]}')
