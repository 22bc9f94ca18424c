#!/bin/bash
# rocprofv3 kernel trace of the bench command; summary copied under gpurun_out/
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${BENCH_ARGS:---steps 3 --warmup 1 --no_cpu_baseline}
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python $REPO/bench.py $ARGS > $REPO/gpurun_out/prof_bench.log 2>&1
STATS=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$STATS" $REPO/gpurun_out/kernel_stats.csv
cat "$STATS"
grep '"metric"' $REPO/gpurun_out/prof_bench.log
