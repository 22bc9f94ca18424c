#!/bin/bash
# Round 4: k_decode_resident with descriptor addressing and branch-free epilogue operand loads.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04i_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04i_pytest.log
tail -6 gpurun_out/r04i_pytest.log
B="--timed device --no_cpu_baseline --no_host_buffers --no_extra_configs"
one() { python bench.py $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['effective']['frac'])"; }
{
for i in 1 2; do
echo "c4      $(one --config 4 --steps 5 --warmup 2)"
echo "U=128   $(one --utterances 128 --steps 5 --warmup 2)"
echo "U=256   $(one --utterances 256 --steps 5 --warmup 2)"
echo "U=65    $(one --utterances 65 --steps 5 --warmup 2)"
echo "c3      $(one --config 3 --steps 3 --warmup 1)"
done
echo "c1 owner select $(one --steps 10 --warmup 3 --flags 2048)"
} 2>&1 | tee gpurun_out/r04i_numbers.txt
timeout 100 python tools/fuzz_gpu.py 70 19 > gpurun_out/r04i_fuzz.txt 2>&1; tail -3 gpurun_out/r04i_fuzz.txt
timeout 120 python tools/stress_persistent.py 2>&1 | tail -3
