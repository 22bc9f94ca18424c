#!/bin/bash
# Round 4: k_decode_resident with per-producer phase words between the dense stages, A/B against the cluster
# barriers (--flags 32768 = UIS_FLAG_CLUSTER_BARRIERS).
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04j_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04j_pytest.log
tail -6 gpurun_out/r04j_pytest.log
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
one() { python bench.py $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['effective']['frac'])"; }
{
for i in 1 2; do
echo "c4 phase words      $(one --config 4 --steps 5 --warmup 2)"
echo "c4 cluster barriers $(one --config 4 --steps 5 --warmup 2 --flags 32768)"
echo "U=128 phase words      $(one --utterances 128 --steps 5 --warmup 2)"
echo "U=128 cluster barriers $(one --utterances 128 --steps 5 --warmup 2 --flags 32768)"
echo "U=256 phase words      $(one --utterances 256 --steps 5 --warmup 2)"
echo "U=256 cluster barriers $(one --utterances 256 --steps 5 --warmup 2 --flags 32768)"
echo "U=65 phase words      $(one --utterances 65 --steps 5 --warmup 2)"
echo "U=65 cluster barriers $(one --utterances 65 --steps 5 --warmup 2 --flags 32768)"
done
} 2>&1 | tee gpurun_out/r04j_numbers.txt
timeout 100 python tools/fuzz_gpu.py 70 23 > gpurun_out/r04j_fuzz.txt 2>&1; tail -2 gpurun_out/r04j_fuzz.txt
timeout 120 python tools/stress_persistent.py 2>&1 | tail -2
timeout 120 python tools/stress_resident.py 60 2>&1 | tail -2
timeout 120 python tools/stream_latency.py 64 300 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k: d[k]['push_1_frame_us_median_c_abi'] for k in ('persistent_launch','one_launch','four_kernels_per_step')})"
