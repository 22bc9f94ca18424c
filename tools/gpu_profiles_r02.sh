#!/bin/bash
# Round-2 evidence, everything under gpurun_out/ (copied to profiles/ afterwards):
#   bench lines of every BASELINE config, rocprofv3 kernel trace of the default bench command,
#   PMC passes, the RCCL path with one rank, per-phase clocks of the one-launch decode,
#   streaming push latency.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
python bench.py > gpurun_out/r02_bench_c1.json 2> gpurun_out/r02_bench_c1.err
for c in 4 3 2; do
  timeout 900 python bench.py --config $c > gpurun_out/r02_bench_c$c.json 2> gpurun_out/r02_bench_c$c.err
done
# the multi-rank code path with one rank: process group, RCCL all_gather, barrier, max-reduce
timeout 600 python bench.py --gpus 1 --force_dist --no_cpu_baseline --no_host_buffers \
  > gpurun_out/r02_force_dist.json 2> gpurun_out/r02_force_dist.err
echo "rc=$?" >> gpurun_out/r02_force_dist.err
# the closed-form model of round 1 on this round's build, for continuity
timeout 600 python bench.py --model tracker --no_cpu_baseline --no_host_buffers > gpurun_out/r02_bench_c1_tracker.json 2>/dev/null
# the launch-per-step path, for comparison
timeout 600 python bench.py --flags 128 --no_cpu_baseline --no_host_buffers > gpurun_out/r02_bench_c1_stepwise.json 2>/dev/null
# kernel trace of the SAME default command
BENCH_ARGS="--steps 10 --warmup 5 --no_cpu_baseline" ./tools/gpu_prof.sh > gpurun_out/r02_prof_stats.log 2>&1
cp gpurun_out/kernel_stats.csv gpurun_out/r02_kernel_stats_bench.csv
grep '"metric"' gpurun_out/prof_bench.log > gpurun_out/r02_bench_line_under_rocprof.json
# PMC passes (separate runs per counter group)
./tools/gpu_pmc.sh > /dev/null 2>&1
cp gpurun_out/pmc.log gpurun_out/r02_pmc_per_kernel.txt
# per-phase clocks (diagnostic build) and streaming latency
if [ -f build/variants/timing.so ]; then bash tools/gpu_timing.sh > /dev/null 2>&1; cp gpurun_out/resident_timing.txt gpurun_out/r02_resident_timing.txt; fi
timeout 300 python tools/stream_latency.py 64 300 2>/dev/null | tail -1 > gpurun_out/r02_stream_latency.json
timeout 300 python tools/stream_latency.py 8 300 2>/dev/null | tail -1 >> gpurun_out/r02_stream_latency.json
head -c 600 gpurun_out/r02_bench_c1.json; echo; for c in 4 3 2; do head -c 300 gpurun_out/r02_bench_c$c.json; echo; done
head -6 gpurun_out/r02_kernel_stats_bench.csv; tail -2 gpurun_out/r02_force_dist.err; head -c 300 gpurun_out/r02_force_dist.json
