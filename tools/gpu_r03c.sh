#!/bin/bash
# per-phase clocks of k_decode_rs variants (diagnostic builds in build/variants/) + quick parity + A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_trained.py -m gpu -x -q 2>&1 | tail -5
for v in ${VARIANTS:-timing}; do
  UIS_LIB_PATH=$PWD/build/variants/$v.so python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_host_buffers --no_extra_configs > gpurun_out/r03c_$v.json 2> gpurun_out/r03c_$v.err
  echo "--- $v"; grep "resident timing" gpurun_out/r03c_$v.err | tail -5; grep -o '"value": [0-9.]*' gpurun_out/r03c_$v.json
done
for i in 1 2; do
  python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_buffers --no_extra_configs > gpurun_out/r03c_bench_rs_$i.json 2> gpurun_out/r03c_bench_rs_$i.err
  python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_buffers --no_extra_configs --flags 2048 > gpurun_out/r03c_bench_owner_$i.json 2> gpurun_out/r03c_bench_owner_$i.err
done
grep -o '"value": [0-9.]*' gpurun_out/r03c_bench_*.json
