#!/bin/bash
# first GPU smoke: parity tests + timing
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
