#!/bin/bash
# Round evidence: bench line + rocprofv3 kernel-trace stats of the SAME command + PMC passes.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python bench.py > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err
grep '"metric"' gpurun_out/bench_default.log
BENCH_ARGS="--steps 10 --warmup 2 --no_cpu_baseline" ./tools/gpu_prof.sh > gpurun_out/prof_stats.log 2>&1
head -12 gpurun_out/kernel_stats.csv
./tools/gpu_pmc.sh > /dev/null 2>&1
grep -A14 "k_decode_resident\|k_dense_input_proj" gpurun_out/pmc.log | head -120
