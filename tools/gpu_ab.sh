#!/bin/bash
# A/B of library variants under build/variants: tests on the default lib, bench on each
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
: > gpurun_out/ab.log
for lib in uisrnn_amd/libuisrnn_hip.so build/variants/*.so; do
  for rep in 1 2; do
    echo "== $lib" >> gpurun_out/ab.log
    UIS_LIB_PATH=$PWD/$lib python bench.py --steps 5 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
  if line.startswith('{'):
    r=json.loads(line); k=r['roofline']['kernel_ms_profile_pass']
    print(r['value'], r['ms_per_step'], 'gru_us', r['roofline']['avg_launch_us'], {a:round(b,2) for a,b in k.items() if b})
" >> gpurun_out/ab.log
  done
done
cat gpurun_out/ab.log
