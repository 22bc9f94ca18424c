#!/bin/bash
# round 3: parity of the paths touched + A/B replicated select (flag 4096) vs owner select + per-phase clocks
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_trained.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03b_tests.txt
cat gpurun_out/r03b_tests.txt
for i in 1 2; do
  python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_buffers --no_extra_configs --flags 4096 > gpurun_out/r03b_bench_rs_$i.json 2> gpurun_out/r03b_bench_rs_$i.err
  python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_buffers --no_extra_configs > gpurun_out/r03b_bench_owner_$i.json 2> gpurun_out/r03b_bench_owner_$i.err
done
grep -o '"value": [0-9.]*' gpurun_out/r03b_bench_*.json
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_host_buffers --no_extra_configs --flags 4096 > gpurun_out/r03b_timing_rs.json 2> gpurun_out/r03b_timing_rs.err
echo "--- RS"; grep "resident timing" gpurun_out/r03b_timing_rs.err | tail -4
tail -3 gpurun_out/r03b_bench_rs_1.err
