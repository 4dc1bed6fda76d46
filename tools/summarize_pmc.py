"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel:
    python tools/summarize_pmc.py <counter_collection.csv> [more.csv ...]
Prints name, dispatches, and per-dispatch mean of every counter."""
import collections
import csv
import re
import sys

for path in sys.argv[1:]:
  acc = collections.defaultdict(lambda: collections.defaultdict(list))
  with open(path, newline='') as f:
    for row in csv.DictReader(f):
      name = re.sub(r'emb::\(anonymous namespace\)::', '', row['Kernel_Name'])
      name = name.split('(')[0][-60:]
      acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
  print(f'# {path}')
  print('kernel,dispatches,counter,mean_per_dispatch')
  for name, counters in sorted(acc.items()):
    for counter, values in counters.items():
      print(f'"{name}",{len(values)},{counter},{sum(values) / len(values):.3f}')
