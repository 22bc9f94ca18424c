#!/bin/bash
# Round-3 evidence, everything under gpurun_out/ (copied to profiles/ afterwards):
#   the default bench line (configs[1] + extra_configs), the owner-select kernels for comparison,
#   a ragged run through the LPT sharder, the RCCL path with one rank, single-utterance latency,
#   rocprofv3 kernel trace of the default bench command, PMC passes, per-phase clocks, a fuzz soak.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
python bench.py > gpurun_out/r03_bench_c1.json 2> gpurun_out/r03_bench_c1.err
# the kernels of round 2 (select of an utterance on one workgroup) on this round's build
python bench.py --flags 2048 --no_cpu_baseline --no_extra_configs > gpurun_out/r03_bench_c1_owner_select.json 2>/dev/null
python bench.py --config 3 --flags 2048 --no_cpu_baseline --no_host_buffers > gpurun_out/r03_bench_c3_owner_select.json 2>/dev/null
python bench.py --config 3 --no_cpu_baseline > gpurun_out/r03_bench_c3.json 2>/dev/null
# ragged utterances (lengths 250..500), dealt by shard_utterances
python bench.py --ragged --no_cpu_baseline --no_extra_configs > gpurun_out/r03_bench_c1_ragged.json 2>/dev/null
# the multi-rank code path with one rank: process group, RCCL all_gather, barrier, max-reduce
timeout 600 python bench.py --gpus 1 --force_dist --no_cpu_baseline --no_host_buffers --no_extra_configs \
  > gpurun_out/r03_force_dist.json 2> gpurun_out/r03_force_dist.err
echo "rc=$?" >> gpurun_out/r03_force_dist.err
# the latency-bound case: ONE utterance of 1000 frames (us per decode step = ms_per_step / 2)
python bench.py --utterances 1 --frames 1000 --no_cpu_baseline --no_extra_configs > gpurun_out/r03_latency_1utt.json 2>/dev/null
python bench.py --utterances 8 --frames 1000 --no_cpu_baseline --no_extra_configs > gpurun_out/r03_latency_8utt.json 2>/dev/null
# kernel trace of the SAME default command (without the extra configs: one kernel population)
BENCH_ARGS="--steps 10 --warmup 3 --no_cpu_baseline --no_extra_configs" ./tools/gpu_prof.sh > gpurun_out/r03_prof_stats.log 2>&1
cp gpurun_out/kernel_stats.csv gpurun_out/r03_kernel_stats_bench.csv
grep '"metric"' gpurun_out/prof_bench.log > gpurun_out/r03_bench_line_under_rocprof.json
# PMC passes (separate runs per counter group)
BENCH_ARGS="--no_extra_configs --no_host_buffers" ./tools/gpu_pmc.sh > /dev/null 2>&1
cp gpurun_out/pmc.log gpurun_out/r03_pmc_per_kernel.txt
# per-phase clocks (diagnostic build)
if [ -f build/variants/timing.so ]; then
  UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_host_buffers --no_extra_configs 2>&1 >/dev/null | grep "resident timing" | tail -4 > gpurun_out/r03_resident_timing.txt
  echo "--- owner-select kernel (k_decode_resident, --flags 2048)" >> gpurun_out/r03_resident_timing.txt
  UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_host_buffers --no_extra_configs --flags 2048 2>&1 >/dev/null | grep "resident timing" | tail -4 >> gpurun_out/r03_resident_timing.txt
  echo "--- k_decode_big<WS> (configs[3] share)" >> gpurun_out/r03_resident_timing.txt
  UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py --config 3 --steps 1 --warmup 0 --no_cpu_baseline --no_host_buffers 2>&1 >/dev/null | grep "resident timing" | tail -4 >> gpurun_out/r03_resident_timing.txt
  echo "--- k_decode_big, owner select (--flags 2048)" >> gpurun_out/r03_resident_timing.txt
  UIS_LIB_PATH=$PWD/build/variants/timing.so python bench.py --config 3 --steps 1 --warmup 0 --no_cpu_baseline --no_host_buffers --flags 2048 2>&1 >/dev/null | grep "resident timing" | tail -4 >> gpurun_out/r03_resident_timing.txt
fi
# soak: random shapes / paths / chunkings against the oracle
timeout 200 python tools/fuzz_gpu.py 120 > gpurun_out/r03_fuzz.txt 2>&1
timeout 300 python tools/stress_resident.py 100 >> gpurun_out/r03_fuzz.txt 2>&1
head -c 700 gpurun_out/r03_bench_c1.json; echo
for f in r03_bench_c1_owner_select r03_bench_c3 r03_bench_c3_owner_select r03_bench_c1_ragged r03_latency_1utt r03_latency_8utt r03_force_dist; do
  python -c "import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; done
head -6 gpurun_out/r03_kernel_stats_bench.csv; cat gpurun_out/r03_resident_timing.txt; tail -3 gpurun_out/r03_fuzz.txt
