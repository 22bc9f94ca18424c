#!/bin/bash
# A/B of library variants in build/variants/ (bench, two interleaved rounds) + per-phase clocks of the timing_* ones
mkdir -p gpurun_out
for v in ${TIMING:-}; do
  UIS_LIB_PATH=$PWD/build/variants/$v.so python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_host_buffers --no_extra_configs > gpurun_out/r03d_$v.json 2> gpurun_out/r03d_$v.err
  echo "--- $v"; grep "resident timing" gpurun_out/r03d_$v.err | tail -4
done
for i in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_buffers --no_extra_configs > gpurun_out/r03d_bench_base_$i.json 2> gpurun_out/r03d_bench_base_$i.err
  for v in ${VARIANTS:-}; do
    UIS_LIB_PATH=$PWD/build/variants/$v.so python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_buffers --no_extra_configs > gpurun_out/r03d_bench_${v}_$i.json 2> gpurun_out/r03d_bench_${v}_$i.err
  done
done
grep -o '"value": [0-9.]*' gpurun_out/r03d_bench_*.json
