#!/bin/bash
# A/B of library variants under build/variants (timing only; variants may be functionally incomplete)
mkdir -p gpurun_out
: > gpurun_out/ab.log
for lib in uisrnn_amd/libuisrnn_hip.so build/variants/*.so; do
    echo "== $lib" >> gpurun_out/ab.log
    UIS_LIB_PATH=$PWD/$lib python bench.py --steps 2 --warmup 1 --no_cpu_baseline --frames 100 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
  if line.startswith('{'):
    r=json.loads(line); k=r['roofline']['kernel_ms_profile_pass']; n=r['roofline']['launches']
    print(r['value'], r['ms_per_step'], {a:round(1e3*b/n,2) for a,b in k.items() if b})
" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
