#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for u in 4 16 64 256 1024; do
    echo "== utterances $u" >> gpurun_out/sweep.log
    python bench.py --steps 2 --warmup 1 --no_cpu_baseline --utterances $u --frames 200 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
  if line.startswith('{'):
    r=json.loads(line); k=r['roofline']['kernel_ms_profile_pass']; n=r['roofline']['launches']
    print(r['value'], r['ms_per_step'], 'rows/launch', r['roofline']['rows_per_launch_executed'], {a:round(1e3*b/n,2) for a,b in k.items() if b})
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
