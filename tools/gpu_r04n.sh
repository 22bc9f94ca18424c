#!/bin/bash
# round 4: k_decode_big<WIN> parity (all GPU tests) + stage clocks of the one-launch window decode
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "look_ahead or calculate_score or golden or probes or python_surface or level" > gpurun_out/r04n_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04n_pytest.log
tail -5 gpurun_out/r04n_pytest.log
UIS_LIB_PATH=$PWD/build/variants/timing.so timeout 300 python bench.py --config 2 --steps 1 --warmup 0 --no_extra_configs --no_cpu_baseline 2>&1 >/dev/null | grep "window launch timing" | tail -4 | tee gpurun_out/r04n_win_timing.txt
