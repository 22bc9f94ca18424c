"""Aggregate rocprofv3 counter_collection CSVs per kernel (mean per dispatch)."""
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
  with open(path) as f:
    for r in csv.DictReader(f):
      name = r['Kernel_Name'].split('(')[0]
      rows[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name in sorted(rows):
  print(name)
  for c in sorted(rows[name]):
    v = rows[name][c]
    print('   %-28s n=%6d mean=%14.1f' % (c, len(v), sum(v) / len(v)))
