#!/bin/bash
# round 3, run A: parity of the replicated-select one-launch decode (k_decode_rs) + A/B against the
# owner-select kernel + per-phase clocks of both
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r03a_tests.txt
cat gpurun_out/r03a_tests.txt
for i in 1 2; do
  python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_buffers --no_extra_configs > gpurun_out/r03a_bench_rs_$i.json 2> gpurun_out/r03a_bench_rs_$i.err
  python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_buffers --no_extra_configs --flags 2048 > gpurun_out/r03a_bench_owner_$i.json 2> gpurun_out/r03a_bench_owner_$i.err
done
grep -o '"value": [0-9.]*' gpurun_out/r03a_bench_*.json
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_host_buffers --no_extra_configs > gpurun_out/r03a_timing_rs.json 2> gpurun_out/r03a_timing_rs.err
UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_host_buffers --no_extra_configs --flags 2048 > gpurun_out/r03a_timing_owner.json 2> gpurun_out/r03a_timing_owner.err
echo "--- RS"; grep "resident timing" gpurun_out/r03a_timing_rs.err | tail -4
echo "--- owner"; grep "resident timing" gpurun_out/r03a_timing_owner.err | tail -4
tail -3 gpurun_out/r03a_bench_rs_1.err
