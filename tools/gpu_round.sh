#!/bin/bash
# tests + smoke + bench + rocprof kernel trace, outputs under gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
python bench.py --steps 5 --warmup 1 > gpurun_out/bench.log 2>&1
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.log
