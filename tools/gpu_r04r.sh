#!/bin/bash
# round 4: per-rank stage clocks of k_decode_big<WIN> (who is the slowest workgroup of a cluster?)
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
UIS_LIB_PATH=$PWD/build/variants/timing.so timeout 300 python bench.py --config 2 --steps 1 --warmup 0 --no_extra_configs --no_cpu_baseline 2>&1 >/dev/null | grep "window launch timing" | tail -10 | tee gpurun_out/r04r_win_timing.txt
