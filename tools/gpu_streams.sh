#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
: > gpurun_out/streams.log
for st in 1 2 4 8; do
  for fl in 0 2; do
    echo "== streams $st flags $fl" >> gpurun_out/streams.log
    python bench.py --steps 5 --warmup 2 --no_cpu_baseline --streams $st --flags $fl 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
  if line.startswith('{'):
    r=json.loads(line); print(r['value'], r['ms_per_step'], r['n_streams'])
" >> gpurun_out/streams.log
  done
done
cat gpurun_out/streams.log
