#!/bin/bash
# round 2: tests + smoke + bench of every BASELINE config, outputs under gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_c1.log 2> gpurun_out/bench_c1.err
for c in 4 3 2; do
  timeout 900 python bench.py --config $c > gpurun_out/bench_c$c.log 2> gpurun_out/bench_c$c.err
done
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; cat gpurun_out/bench_c*.log; tail -3 gpurun_out/bench_c*.err
