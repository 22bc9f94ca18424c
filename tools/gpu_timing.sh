#!/bin/bash
# per-phase wall-clock table of the one-launch decode (diagnostic build -DUIS_RESIDENT_TIMING in
# build/variants/timing.so), bench workload; output: gpurun_out/resident_timing.txt
mkdir -p gpurun_out
UIS_LIB_PATH=$PWD/build/variants/timing.so timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_host_buffers $ARGS \
  > gpurun_out/resident_timing.json 2> gpurun_out/resident_timing.err
grep "resident timing" gpurun_out/resident_timing.err | tail -6 > gpurun_out/resident_timing.txt
cat gpurun_out/resident_timing.txt; grep -o '"value": [0-9.]*' gpurun_out/resident_timing.json
