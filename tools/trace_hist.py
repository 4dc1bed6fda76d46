import csv, sys, statistics, collections
path, needle = sys.argv[1], sys.argv[2]
d=[]
with open(path, newline='') as f:
  for row in csv.DictReader(f):
    if needle in row['Kernel_Name']:
      d.append((int(row['Start_Timestamp']), int(row['End_Timestamp'])-int(row['Start_Timestamp'])))
d.sort()
dur=[x[1] for x in d]
n=len(dur)
print(needle, 'n', n, 'mean', round(statistics.mean(dur)), 'median', statistics.median(dur))
for a,b in ((0,n//5),(n//5,2*n//5),(2*n//5,3*n//5),(3*n//5,4*n//5),(4*n//5,n)):
  seg=dur[a:b]; print('  fifth', a, 'median', statistics.median(seg), 'mean', round(statistics.mean(seg)))
h=collections.Counter(x//2000*2 for x in dur)
print('  hist us:', sorted(h.items()))
