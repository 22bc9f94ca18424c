#!/bin/bash
# A/B of library variants under build/variants against the in-tree library, interleaved, on one box
# (timing only; variants may be diagnostic builds).  ARGS: extra bench.py arguments.
mkdir -p gpurun_out
: > gpurun_out/ab.log
for rep in ${REPS:-1 2 3}; do
  for lib in uisrnn_amd/libuisrnn_hip.so build/variants/*.so; do
    v=$(UIS_LIB_PATH=$PWD/$lib timeout 120 python bench.py --steps 5 --warmup 1 --no_cpu_baseline $ARGS 2>/dev/null | grep -o '"value": [0-9.]*')
    echo "rep=$rep $lib $v" >> gpurun_out/ab.log
  done
done
sort -k2 gpurun_out/ab.log
