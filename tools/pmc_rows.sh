#!/bin/bash
# per-dispatch FETCH_SIZE / WRITE_SIZE rows of the decode kernel (one bench sub-run each): tools/pmc_rows.sh [bench args]
cd /tmp; export TMPDIR=/tmp UIS_BENCH_CHILD=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pr; (ulimit -c 0; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pr -o p -- python $GRAFT_REPO_ROOT/bench.py --timed device --steps 2 --warmup 0 --no_cpu_baseline --no_host_buffers --no_extra_configs "$@" > /dev/null 2>&1)
  python - $c <<'PY'
import csv, glob, sys
for path in glob.glob('/tmp/pr/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(path)):
    if 'k_decode' in r['Kernel_Name'] and r['Counter_Name'] == sys.argv[1]:
      print(sys.argv[1], r['Dispatch_Id'], r['Kernel_Name'][:40], r['Counter_Value'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 'ms')
PY
done
